"""DIAGNOSTIC: lanes=states scan kernels against the generic kernels on the same inputs (GPU only).
usage: python tools/ls_diff.py [batch dim N L G dtype]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vivim_amd import _lib, selective_scan_cuda as ss

def run(batch, dim, N, L, G, dtype, fwd, bwd, t):
    Lb = _lib.lib()
    Lb.vivim_set_tuning(0, fwd); Lb.vivim_set_tuning(1, bwd)
    res = ss.fwd(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], t["z"], t["bias"], True)
    out, x, out_z = res[0], res[1], res[2]
    g = ss.bwd(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], t["z"], t["bias"], t["dout"], x, out, None, True, False)
    torch.cuda.synchronize()
    return dict(out=out, out_z=out_z, last=x[:, :, -1], du=g[0], ddelta=g[1], dA=g[2], dB=g[3], dC=g[4], dD=g[5], dbias=g[6], dz=g[7])

def main():
    a = sys.argv[1:]
    batch, dim, N, L, G = (int(v) for v in a[:5]) if len(a) >= 5 else (1, 8, 16, 1, 1)
    dtype = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[a[5] if len(a) > 5 else "fp32"]
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=gen)
    t = dict(u=r(batch, dim, L).to(dev, dtype), delta=(0.5 * torch.rand(batch, dim, L, generator=gen)).to(dev, dtype),
             A=(-0.5 * torch.rand(dim, N, generator=gen) - 0.05).to(dev), B=r(batch, G, N, L).to(dev, dtype), C=r(batch, G, N, L).to(dev, dtype),
             D=r(dim).to(dev), z=r(batch, dim, L).to(dev, dtype), bias=(0.5 * torch.rand(dim, generator=gen)).to(dev),
             dout=r(batch, dim, L).to(dev, dtype))
    ref = run(batch, dim, N, L, G, dtype, 3, 3, t)
    new = run(batch, dim, N, L, G, dtype, 6, 4, t)
    for k in ref:
        a_, b_ = new[k].float(), ref[k].float()
        err = (a_ - b_).norm() / b_.norm().clamp_min(1e-30)
        line = f"{k:7s} rel {err:.3e}  max|d| {(a_ - b_).abs().max():.3e}"
        if err > 1e-3 and a_.dim() >= 2:
            d = (a_ - b_).abs()
            per_ch = d.amax(dim=tuple(i for i in range(d.dim()) if i != 1)) if k not in ("dA",) else d.amax(dim=1)
            line += "  bad channel-axis idx: " + str([int(i) for i in torch.nonzero(per_ch > 1e-3 * b_.abs().max()).flatten()[:24]])
            if d.dim() == 3:
                per_t = d.amax(dim=(0, 1))
                line += "  bad t: " + str([int(i) for i in torch.nonzero(per_t > 1e-3 * b_.abs().max()).flatten()[:24]])
        print(line)

main()
