#!/usr/bin/env python3
"""Is the v3 Mamba module itself host-bound?  Wall time per forward+backward of one module on each stage shape of config 2
(B 3, nf 5, 256x256, bf16 autocast) against the GPU-busy time of the same loop (sum of kernel durations: run this script
under `rocprofv3 --kernel-trace --stats` and divide the total by the iterations printed here)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mamba_ssm import Mamba

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
torch.manual_seed(0)
for st, (dim, stride) in enumerate(zip((64, 128, 320, 512), (4, 8, 16, 32))):
    L = 5 * (256 // stride) ** 2
    m = Mamba(d_model=dim, d_state=16, expand=2, bimamba_type="v3").to(dev)
    x = torch.randn(3, L, dim, device=dev, requires_grad=True)

    def it():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x, nframes=5)
        y.float().square().mean().backward()

    for _ in range(5):
        it()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(iters):
        it()
    host = time.perf_counter() - t0          # the host is done issuing here
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"stage {st} dim {dim} L {L}: host issue {host / iters * 1e3:.3f} ms/iter, wall {wall / iters * 1e3:.3f} ms/iter, "
          f"gpu span {e0.elapsed_time(e1) / iters:.3f} ms/iter ({iters} iters + 5 warm-up)", flush=True)
