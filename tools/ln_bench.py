#!/usr/bin/env python3
"""csrc/layernorm.hip against ATen's layer_norm on MambaLayer's channel-major view, forward + backward, at the four stage shapes
of BASELINE configs[1] (B 3, 5 frames, f32 residual stream under bf16 autocast)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vivim_amd import layernorm as ln  # noqa: E402


def timeit(fn, iters=50, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


dev = torch.device("cuda:0")
only = [int(a) for a in sys.argv[1:]]            # stages to run (default: all four)
for st, (C, stride) in enumerate(zip((64, 128, 320, 512), (4, 8, 16, 32))):
    if only and st not in only:
        continue
    B, L = 3, 5 * (256 // stride) ** 2
    x = torch.randn(B, C, L, device=dev).transpose(1, 2).requires_grad_(True)
    w, b = torch.ones(C, device=dev, requires_grad=True), torch.zeros(C, device=dev, requires_grad=True)
    g = torch.randn(B, L, C, device=dev)

    def aten():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = F.layer_norm(x, (C,), w, b, 1e-5)
        y.backward(g)

    def fused():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = ln.layer_norm_cm(x, w, b, 1e-5)
        y.backward(g)
    ta, tf = timeit(aten), timeit(fused)
    nbytes = 5 * B * L * C * 4            # x, y; dy, x, dx
    print(f"stage {st} C={C} L={L}: ATen {ta:7.1f} us   fused {tf:7.1f} us   ({nbytes / tf / 1e3:7.1f} GB/s algorithmic, fwd + bwd)", flush=True)
