#!/usr/bin/env python3
"""One MambaLayer (modeling/vivim.py:111-159) forward + backward at a stage shape of BASELINE configs[1] (B 3, 5 frames, bf16
autocast), a few iterations: run under rocprofv3 (tools/layer_prof.sh) it lists what the layer costs on the GPU, kernel by kernel.
    python tools/layer_prof.py <stage 0..3> [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modeling.vivim import MambaLayer  # noqa: E402

st = int(sys.argv[1])
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
C, stride = ((64, 4), (128, 8), (320, 16), (512, 32))[st]
dev = torch.device("cuda:0")
torch.manual_seed(0)
layer = MambaLayer(C).to(dev)
x = torch.randn(3, C, 5, 256 // stride, 256 // stride, device=dev, requires_grad=True)
g = torch.randn_like(x)
for _ in range(iters + 2):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = layer(x)
    y.backward(g)
torch.cuda.synchronize()
print("stage", st, "C", C, "tokens", 3 * x.shape[2] * x.shape[3] * x.shape[4], "iterations", iters + 2)
