#!/usr/bin/env python3
"""Host-side profile of the benchmark's train step (cProfile, main thread: forward + loss + optimizer; the
autograd engine runs custom backward functions on its own thread, which cProfile does not see).
    python tools/host_profile.py [steps]  ->  top functions by own time / cumulative time."""
import cProfile, io, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vivim_amd.train_step import build_model, make_optimizer, synthetic_batch, train_step

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
torch.manual_seed(42)
model = build_model(3, dev, mamba_kwargs={"d_state": 16, "expand": 2})
clip, onehot = synthetic_batch(3, 5, 256, 3, dev, 42)
opt = make_optimizer(model)
for _ in range(3):
    train_step(model, opt, clip, onehot, 3, torch.bfloat16)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    train_step(model, opt, clip, onehot, 3, torch.bfloat16)
torch.cuda.synchronize()
print(f"unprofiled: {(time.perf_counter() - t0) / steps * 1e3:.1f} ms/step")
# forward only, to split host time between forward and backward
with torch.no_grad():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            model(clip)
    torch.cuda.synchronize()
print(f"forward only (no_grad): {(time.perf_counter() - t0) / steps * 1e3:.1f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    train_step(model, opt, clip, onehot, 3, torch.bfloat16)
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(28)
    print(f"==== by {key} ({steps} steps) ====")
    print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000])
