#!/usr/bin/env python3
"""Kernel micro-benchmark: times the four hot-path kernels (through the C ABI) on the stage shapes of a
BASELINE.json config and prints achieved algorithmic GB/s (SURVEY.md section 8d formulas).
Usage: python tools/kbench.py [--config 2|3|5] [--iters 20] [--stages 0,1,2,3] [--kernels sf,sb,cf,cb] [--groups 3]
--groups 3: the grouped v3 shapes (three directions side by side: dim = 3 * d_inner, n_groups = 3, contiguous rows).
--cold S: cycle S independent input sets (S >= 3 and S * set size > the 256 MB Infinity Cache: no launch finds its inputs
in a cache).  --ceiling: also print the device-to-device copy rate (the practical HBM ceiling)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import causal_conv1d_cuda as cc  # noqa: E402
import selective_scan_cuda as ss  # noqa: E402

CONFIGS = {  # batch, nf, image, N, expand, dtype
    2: (3, 5, 256, 16, 2, torch.bfloat16),
    3: (8, 5, 512, 16, 2, torch.float32),
    5: (1, 8, 256, 64, 4, torch.bfloat16),
}
DIMS, STRIDES = [64, 128, 320, 512], [4, 8, 16, 32]


def alg_bytes(kind, B, D, L, N, s, G=1, W=4):
    if kind == "sf":
        return 5 * B * D * L * s + 2 * B * G * N * L * s + 4 * (D * N + 2 * D)
    if kind == "sb":   # out_z not requested: 8 activation streams
        return 8 * B * D * L * s + 2 * B * G * N * L * s + 2 * B * G * N * L * 4 + 4 * (2 * D * N + 4 * D)
    if kind == "cf":
        return 2 * B * D * L * s + 4 * D * (W + 1)
    return 3 * B * D * L * s + 8 * D * (W + 1)


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--stages", default="0,1,2,3")
    ap.add_argument("--kernels", default="sf,sb,cf,cb")
    ap.add_argument("--groups", type=int, default=1)
    ap.add_argument("--cold", type=int, default=1)
    ap.add_argument("--ceiling", action="store_true")
    ap.add_argument("--pad", type=int, default=0, help="extra elements between the rows of the inputs (row stride L + pad)")
    a = ap.parse_args()
    B, nf, img, N, expand, dt = CONFIGS[a.config]
    s = torch.finfo(dt).bits // 8
    dev = torch.device("cuda:0")
    print(f"config {a.config}: B={B} nf={nf} img={img} N={N} expand={expand} dtype={dt}" + (f" cold x{a.cold}" if a.cold > 1 else ""))
    if a.ceiling:
        src = torch.ones(1 << 30, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        t = timeit(lambda: dst.copy_(src), 10)
        print(f"  device-to-device copy of 1 GiB: {2 * src.numel() / t / 1e9:8.1f} GB/s (read + write)")
        del src, dst
    for st in map(int, a.stages.split(",")):
        G = a.groups
        D = DIMS[st] * expand * G
        L = nf * (img // STRIDES[st]) ** 2
        P = a.pad
        mk = lambda *sh: torch.randn(*sh[:-1], sh[-1] + P, device=dev).to(dt)[..., :sh[-1]]
        strided = (lambda: mk(D, B, L).transpose(0, 1)) if G == 1 else (lambda: mk(B, D, L))
        A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(D, 1)
        Dv, bias = torch.ones(D, device=dev), torch.full((D,), -4.0, device=dev)
        w, cb = torch.randn(D, 4, device=dev), torch.randn(D, device=dev)
        sets = []
        for _ in range(max(1, a.cold)):
            u, z, dout = strided(), strided(), strided()
            delta = 0.2 * strided()
            Bm, Cm = mk(B, G, N, L), mk(B, G, N, L)
            out, x, out_z = ss.fwd(u, delta, A, Bm, Cm, Dv, z, bias, True)
            sets.append((u, delta, z, dout, Bm, Cm, out, x, torch.empty_like(z)))
        turn = [0]

        def nxt():
            turn[0] = (turn[0] + 1) % len(sets)
            return sets[turn[0]]

        def sf():
            u, delta, z, dout, Bm, Cm, out, x, dz = nxt()
            return ss.fwd(u, delta, A, Bm, Cm, Dv, z, bias, True)

        def sb():
            u, delta, z, dout, Bm, Cm, out, x, dz = nxt()
            return ss.bwd(u, delta, A, Bm, Cm, Dv, z, bias, dout, x, out, dz, True, False)

        def cf():
            return cc.causal_conv1d_fwd(nxt()[0], w, cb, True)

        def cbw():
            s_ = nxt()
            return cc.causal_conv1d_bwd(s_[0], w, cb, s_[3], None, True)

        runs = {"sf": sf, "sb": sb, "cf": cf, "cb": cbw}
        for k in a.kernels.split(","):
            t = timeit(runs[k], a.iters)
            nb = alg_bytes(k, B, D, L, N, s, G)
            cut = ""
            if k == "sb":      # segments of the token axis, from the workspace the backward asked for
                cut = f"  S={ss.last_workspace_bytes.get('bwd', 0) // (B * D * (2 * N + 1) * 4) or 1}"
            print(f"  stage {st} D={D:5d} L={L:6d} {k}: {t * 1e6:9.1f} us  {nb / 1e6:8.1f} MB  {nb / t / 1e9:8.1f} GB/s "
                  f"({nb / t / 8e12 * 100:5.1f}% of 8 TB/s){cut}", flush=True)


if __name__ == "__main__":
    main()
