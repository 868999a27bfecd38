#!/usr/bin/env python3
"""Host time of one MambaLayer forward + backward (stage shape of configs[1], bf16 autocast): cProfile, this package's functions
by internal time.   python tools/layer_host.py <stage> [iters]"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modeling.vivim import MambaLayer  # noqa: E402

st = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
C, stride = ((64, 4), (128, 8), (320, 16), (512, 32))[st]
dev = torch.device("cuda:0")
torch.manual_seed(0)
layer = MambaLayer(C).to(dev)
x = torch.randn(3, C, 5, 256 // stride, 256 // stride, device=dev, requires_grad=True)
g = torch.randn_like(x)


def it():
    for p in layer.parameters():
        p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = layer(x)
    y.backward(g)


for _ in range(10):
    it()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    it()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"stage {st}: host {1e6 * (t1 - t0) / iters:.0f} us per forward + backward submitted, {1e6 * (t2 - t0) / iters:.0f} us with the queue drained")
pr = cProfile.Profile()
pr.enable()
for _ in range(iters):
    it()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(22)
    print(f"==== per {iters} iterations, by {key}")
    print("\n".join(l[:150] for l in s.getvalue().splitlines() if l.strip())[-4500:])
