#!/usr/bin/env python3
"""Host time per call of the fused LayerNorm wrapper against ATen layer_norm (the step is host-paced): submit time and time with the queue drained."""
import os, sys, time, cProfile, pstats
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vivim_amd import layernorm as ln
dev = torch.device("cuda:0")
C, L, B = 128, 5120, 3
x = torch.randn(B, C, L, device=dev).transpose(1, 2).requires_grad_(True)
w, b = torch.ones(C, device=dev, requires_grad=True), torch.zeros(C, device=dev, requires_grad=True)
g = torch.randn(B, L, C, device=dev)
def host(fn, n=300):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t) / n * 1e6, (t2 - t) / n * 1e6
def f_fwd():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return ln.layer_norm_cm(x, w, b, 1e-5)
def a_fwd():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return F.layer_norm(x, (C,), w, b, 1e-5)
def f_all(): f_fwd().backward(g)
def a_all(): a_fwd().backward(g)
def f_nograd():
    with torch.no_grad(): return f_fwd()
for name, fn in (("fused fwd", f_fwd), ("aten fwd", a_fwd), ("fused fwd nograd", f_nograd), ("fused all", f_all), ("aten all", a_all)):
    print(name, "host %.1f us, with drain %.1f us" % host(fn), flush=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(300): f_all()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
