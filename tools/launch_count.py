"""Kernel launches of one tri-directional Mamba block (forward + backward, bf16 autocast, stage-0 shape of the bench) and of
one whole train step, by kernel name -- the step is host-bound, so launches are what it pays for.
    python tools/launch_count.py [--dim 64] [--batch 3] [--frames 5] [--hw 4096] [--step]"""
import argparse
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def launches(fn, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        fn()
        torch.cuda.synchronize()
    names = collections.Counter()
    dur = collections.Counter()
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA:
            names[ev.name[:90]] += 1
            dur[ev.name[:90]] += ev.device_time
    return names, dur


def show(title, names, dur, top=40):
    print(f"== {title}: {sum(names.values())} launches, {sum(dur.values()) / 1e3:.2f} ms of GPU time")
    for n, c in names.most_common(top):
        print(f"   {c:5d} {dur[n] / 1e3:8.3f} ms  {n}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--batch", type=int, default=3)
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--hw", type=int, default=4096)
    ap.add_argument("--step", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    from vivim_amd.mamba_simple import Mamba
    torch.manual_seed(0)
    m = Mamba(d_model=a.dim, bimamba_type="v3", nframes=a.frames).to(dev)
    x = torch.randn(a.batch, a.frames * a.hw, a.dim, device=dev, requires_grad=True)

    def block():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x, nframes=a.frames)
        y.float().square().mean().backward()
    show(f"Mamba v3 block dim {a.dim} L {a.frames * a.hw} batch {a.batch}", *launches(block))
    if a.step:
        from vivim_amd import train_step as ts
        model = ts.build_model(3, dev)
        opt = ts.make_optimizer(model)
        clip, onehot = ts.synthetic_batch(a.batch, a.frames, 256, 3, dev, 0)

        def step():
            ts.train_step(model, opt, clip, onehot, 3, torch.bfloat16)
        show("train step", *launches(step), top=60)


if __name__ == "__main__":
    main()
