"""Where the host time of a train step goes (the step is host-paced): cProfile over a few steps, the vivim_amd wrappers and the
top of the list.   python tools/host_profile.py [steps]"""
import cProfile, io, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vivim_amd import train_step as ts

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ts.build_model(3, dev)
opt = ts.make_optimizer(model)
clip, onehot = ts.synthetic_batch(3, 5, 256, 3, dev, 0)
for _ in range(3):
    ts.train_step(model, opt, clip, onehot, 3, torch.bfloat16)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    ts.train_step(model, opt, clip, onehot, 3, torch.bfloat16)
torch.cuda.synchronize()
pr.disable()
for key, pat in (("cumulative", "vivim_amd"), ("tottime", None)):
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s).sort_stats(key)
    st.print_stats(pat, 30) if pat else st.print_stats(25)
    print(f"==== per {steps} steps, sorted by {key}" + (f", filter {pat}" if pat else ""))
    print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000])
