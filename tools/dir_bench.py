#!/usr/bin/env python3
"""Times the two direction-map kernels (vivim_dir_scatter / vivim_dir_gather) and, for scale, the grouped conv1d + scan
forward and backward that consume their output, at the four stage shapes of BASELINE configs[1] (bf16, B 3, 5 frames).
With VIVIM_LIB pointing at the -DDIR_ABL=1 build (tools/abl.sh dirbuild) the maps move the frame interleave only: the
difference is the most a `direction` argument on conv + scan (directions 0 and 1) could save."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vivim_amd import dirmap  # noqa: E402


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
B, nf, dt = 3, 5, torch.bfloat16
tot_s = tot_g = 0.0
for st, (dim, stride) in enumerate(zip((64, 128, 320, 512), (4, 8, 16, 32))):
    D, L = 2 * dim, nf * (256 // stride) ** 2
    xz = torch.randn(2 * D, B, L, device=dev).to(dt).transpose(0, 1)          # (B, 2D, L), strides (L, B*L, 1)
    o3 = torch.randn(B, 3, D, L, device=dev).to(dt)
    ts = timeit(lambda: dirmap._scatter(xz, nf, D, 1.0))
    tg = timeit(lambda: dirmap._gather(o3.view(B, 1, 3, D, L), nf, 1.0 / 3.0))
    tg2 = timeit(lambda: dirmap._gather(dirmap._scatter(xz, nf, D, 1.0), nf, 1.0))   # backward of stack_directions: (B, 2, 3, D, L)
    tot_s += ts
    tot_g += tg
    print(f"stage {st} D={D} L={L}: scatter xz {ts:7.1f} us ({4 * B * 2 * D * L * 2 / ts / 1e3:7.1f} GB/s)   gather out {tg:7.1f} us", flush=True)
print(f"per layer forward: scatter + gather, four stages summed: {tot_s + tot_g:.1f} us")
