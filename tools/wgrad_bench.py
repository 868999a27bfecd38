#!/usr/bin/env python3
"""csrc/wgrad.hip against torch.bmm on the weight-gradient products of the grouped op's backward at the four stage shapes of
BASELINE configs[1] (B 3, 5 frames, three directions, bf16): run under rocprofv3 for kernel durations (tools/wgrad_prof.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vivim_amd import wgrad  # noqa: E402

dev = torch.device("cuda:0")
only = [int(a) for a in sys.argv[1:]]
for st, (C, stride) in enumerate(zip((64, 128, 320, 512), (4, 8, 16, 32))):
    if only and st not in only:
        continue
    D, R, N, K = 2 * C, (C + 15) // 16, 16, 3 * 5 * (256 // stride) ** 2
    ddelta = torch.randn(3, D, K, device=dev).to(torch.bfloat16)
    x_dbl = torch.randn(3, R + 2 * N, K, device=dev).to(torch.bfloat16)
    conv = torch.randn(3, D, K, device=dev).to(torch.bfloat16)
    for _ in range(20):
        wgrad.wgrad_nt(ddelta, x_dbl[:, :R])
        wgrad.wgrad_nt(x_dbl, conv)
        torch.bmm(ddelta, x_dbl[:, :R].transpose(1, 2))
        torch.bmm(x_dbl, conv.transpose(1, 2))
    torch.cuda.synchronize()
    print(f"stage {st}: ddelta_proj_weight ({D} x {R}), dx_proj_weight ({R + 2 * N} x {D}), K = {K}")
