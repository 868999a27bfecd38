#!/usr/bin/env python3
"""Debug aid: run one golden scan fixture through fwd + bwd and print the error of every output."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import selective_scan_cuda as ss
name = sys.argv[1] if len(sys.argv) > 1 else "scan_n64"
g = dict(np.load(os.path.join(ROOT, "tests/golden", name + ".npz")))
dev = torch.device("cuda")
T = lambda k: torch.from_numpy(g[k]).to(dev) if k in g else None
u, delta, A, B, C, D, z, bias, dout = (T(k) for k in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias", "dout"))
meta = g["meta"]; print("meta", meta, "dtype", g["dtype"])
softplus = True
for rep in range(2):
    res = ss.fwd(u, delta, A, B, C, D, z, bias, softplus)
    out, x = res[0], res[1]
    dz = torch.empty_like(z) if z is not None else None
    grads = ss.bwd(u, delta, A, B, C, D, z, bias, dout, x, out if z is not None else None, dz, softplus, False)
    names = ["du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias", "dz"]
    for n, t in zip(names, grads):
        if t is None or n not in g: continue
        r = torch.from_numpy(g[n]).to(dev).float()
        e = (t.float() - r).abs().max().item() / (r.abs().max().item() + 1e-30)
        bad = (~torch.isfinite(t.float())).sum().item()
        print(f"rep{rep} {n:12s} rel_err {e:.3e} nonfinite {bad} shape {tuple(t.shape)}")
    if rep == 0 and (grads[0].float() - torch.from_numpy(g['du']).to(dev)).abs().max() > 1e-2:
        print("du[0,:,:8] ", grads[0][0, :, :8].cpu().numpy())
        print("ref        ", g["du"][0, :, :8])
