#!/usr/bin/env python3
"""Inference frames/s of the Vivim model (the reference's inference.py:294-325, 414 times a
no_grad forward per clip without a device sync; here the timed region is bracketed by torch.cuda.synchronize()).
    python tools/infer_fps.py [--batch 1] [--clip-length 5] [--image-size 256] [--dtype bf16] [--iters 30]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vivim_amd.train_step import build_model, synthetic_batch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--clip-length", type=int, default=5)
ap.add_argument("--image-size", type=int, default=256)
ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda", 0)
amp = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[a.dtype]
torch.manual_seed(42)
model = build_model(3, dev, mamba_kwargs={"d_state": 16, "expand": 2}).eval()
clip, _ = synthetic_batch(a.batch, a.clip_length, a.image_size, 3, dev, 42)
with torch.no_grad(), torch.autocast("cuda", dtype=amp, enabled=amp != torch.float32):
    for _ in range(5):
        model(clip)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        out = model(clip)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
print(json.dumps({"metric": "inference frames/sec (no_grad forward, eval)", "value": round(a.batch * a.clip_length / dt, 2),
                  "ms_per_clip_batch": round(dt * 1e3, 3), "batch": a.batch, "clip_length": a.clip_length,
                  "image_size": a.image_size, "dtype": a.dtype, "finite": bool(torch.isfinite(out).all())}))
