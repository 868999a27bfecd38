#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference's OWN pure-PyTorch oracles.

Runs only in the build container (needs /root/reference); the GPU box and the test-suite read the
committed .npz files.  The reference's CUDA extension modules do not exist here, so the two names it
imports at module scope (`causal_conv1d_cuda`, `selective_scan_cuda`; selective_scan_interface.py:9-11,
causal_conv1d_interface.py:7) are registered as empty modules and the interface files are loaded by
path (bypassing mamba_ssm/__init__.py, which drags in the LM stack).  Only functions that never touch
those extensions are called:
  selective_scan_ref, causal_conv1d_ref, mamba_inner_ref (with its two fn names rebound to the refs),
  and Mamba.forward's v3 branch with mamba_inner_fn_no_out_proj rebound to a composition of the refs.
Gradients come from torch autograd through those refs, exactly as the reference's tests obtain them
(mamba/tests/ops/test_selective_scan.py:121-149, causal-conv1d/tests/test_causal_conv1d.py:62-75).

Usage:  python tests/golden/make_golden.py   (rewrites every fixture deterministically)
"""
import importlib.util
import os
import sys
import types
import warnings

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
warnings.filterwarnings("ignore", category=FutureWarning)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    for n in ("causal_conv1d_cuda", "selective_scan_cuda"):
        sys.modules.setdefault(n, types.ModuleType(n))
    cci = _load("causal_conv1d.causal_conv1d_interface",
                f"{REF}/causal-conv1d/causal_conv1d/causal_conv1d_interface.py")
    pkg = types.ModuleType("causal_conv1d")
    pkg.causal_conv1d_fn = cci.causal_conv1d_fn
    pkg.causal_conv1d_update = cci.causal_conv1d_update
    sys.modules["causal_conv1d"] = pkg
    for n in ("mamba_ssm", "mamba_ssm.ops", "mamba_ssm.modules"):
        sys.modules.setdefault(n, types.ModuleType(n))
    ssi = _load("mamba_ssm.ops.selective_scan_interface",
                f"{REF}/mamba/mamba_ssm/ops/selective_scan_interface.py")
    # mamba_inner_ref calls the CUDA-backed fns by module-global name; point them at the refs
    ssi.causal_conv1d_fn = cci.causal_conv1d_ref
    ssi.selective_scan_fn = ssi.selective_scan_ref
    return cci, ssi


def gen(seed):
    return torch.Generator().manual_seed(seed)


def cast(t, dtype):
    return t.to(dtype)


def npf(t):
    return None if t is None else t.detach().float().numpy()


def save(name, **arrays):
    arrays = {k: v for k, v in arrays.items() if v is not None}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    print(f"wrote {name}.npz  ({sum(a.nbytes for a in arrays.values() if hasattr(a, 'nbytes')) / 1e3:.1f} kB raw)")


DT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}

# name, batch, dim, dstate, seqlen, groups, varB, varC, hasD, hasZ, hasBias, softplus, dtype, init
SCAN_CASES = [
    ("scan_tss_l128",   2, 4, 8, 128, 1, 1, 1, 1, 1, 1, 1, "fp32", "test"),   # test_selective_scan.py:54-56
    ("scan_odd_l151_g2", 2, 4, 8, 151, 2, 1, 1, 1, 1, 1, 1, "fp32", "test"),
    ("scan_l1",         1, 8, 16, 1, 1, 1, 1, 1, 1, 1, 1, "fp32", "test"),
    ("scan_l7_plain",   1, 6, 16, 7, 1, 1, 1, 0, 0, 0, 0, "fp32", "test"),
    ("scan_stage3",     2, 64, 16, 320, 1, 1, 1, 1, 1, 1, 1, "fp32", "module"),
    ("scan_n64",        1, 4, 64, 64, 1, 1, 1, 1, 1, 1, 1, "fp32", "module"),
    ("scan_l2049",      1, 4, 8, 2049, 1, 1, 1, 1, 1, 1, 1, "fp32", "test"),  # crosses the ref's 2048 chunk
    ("scan_constBC",    2, 4, 8, 64, 1, 0, 0, 1, 1, 1, 1, "fp32", "test"),
    ("scan_bf16_l372",  2, 4, 8, 372, 1, 1, 1, 1, 1, 1, 1, "bf16", "test"),
    ("scan_fp16_l256",  2, 4, 8, 256, 2, 1, 1, 1, 1, 1, 1, "fp16", "test"),
    ("scan_noz_l1024",  1, 4, 16, 1024, 1, 1, 1, 1, 0, 1, 1, "fp32", "module"),
]


def make_scan(ssi):
    for i, (name, b, d, n, l, g, vB, vC, hD, hZ, hb, sp, dt, init) in enumerate(SCAN_CASES):
        G = gen(100 + i)
        dtype = DT[dt]
        if init == "test":   # distributions of test_selective_scan.py:58-88
            A = -0.5 * torch.rand(d, n, generator=G)
            delta = 0.5 * torch.rand(b, d, l, generator=G)
            bias = 0.5 * torch.rand(d, generator=G)
        else:                # module init, mamba_simple.py:99-117
            A = -torch.arange(1, n + 1, dtype=torch.float32).repeat(d, 1)
            delta = 0.2 * torch.randn(b, d, l, generator=G)
            dtv = torch.exp(torch.rand(d, generator=G) * (np.log(0.1) - np.log(1e-3)) + np.log(1e-3))
            bias = dtv + torch.log(-torch.expm1(-dtv))
        u = torch.randn(b, d, l, generator=G)
        Bm = torch.randn(b, g, n, l, generator=G) if vB else torch.randn(d, n, generator=G)
        Cm = torch.randn(b, g, n, l, generator=G) if vC else torch.randn(d, n, generator=G)
        D = torch.randn(d, generator=G) if hD else None
        z = torch.randn(b, d, l, generator=G) if hZ else None
        bias = bias if hb else None
        dout = torch.randn(b, d, l, generator=G)
        u, delta, dout = (cast(t, dtype) for t in (u, delta, dout))
        z = cast(z, dtype) if z is not None else None
        if vB:
            Bm = cast(Bm, dtype)
        if vC:
            Cm = cast(Cm, dtype)
        leaves = [t for t in (u, delta, A, Bm, Cm, D, z, bias) if t is not None]
        for t in leaves:
            t.requires_grad_(True)
        out, last = ssi.selective_scan_ref(u, delta, A, Bm, Cm, D, z=z, delta_bias=bias,
                                           delta_softplus=bool(sp), return_last_state=True)
        out.backward(dout)
        save(name, u=npf(u), delta=npf(delta), A=npf(A), B=npf(Bm), C=npf(Cm), D=npf(D), z=npf(z),
             delta_bias=npf(bias), dout=npf(dout), out=npf(out), last_state=npf(last),
             du=npf(u.grad), ddelta=npf(delta.grad), dA=npf(A.grad), dB=npf(Bm.grad), dC=npf(Cm.grad),
             dD=npf(D.grad) if D is not None else None, dz=npf(z.grad) if z is not None else None,
             ddelta_bias=npf(bias.grad) if bias is not None else None,
             meta=np.array([b, d, n, l, g, vB, vC, hD, hZ, hb, sp]), dtype=np.array(dt))


# name, batch, dim, seqlen, width, silu, bias, dtype
CONV_CASES = [
    ("conv_w4_silu_l151", 2, 8, 151, 4, 1, 1, "fp32"),
    ("conv_w3_l7",        1, 8, 7, 3, 0, 1, "fp32"),
    ("conv_w2_nobias_l1", 2, 8, 1, 2, 1, 0, "fp32"),
    ("conv_w4_silu_l1134", 2, 8, 1134, 4, 1, 1, "fp32"),   # odd length from test_causal_conv1d.py:24-25
    ("conv_w4_bf16_l372", 2, 8, 372, 4, 1, 1, "bf16"),
    ("conv_w4_fp16_l64",  2, 8, 64, 4, 0, 0, "fp16"),
]


def make_conv(cci):
    for i, (name, b, d, l, w, silu, hb, dt) in enumerate(CONV_CASES):
        G = gen(200 + i)
        dtype = DT[dt]
        x = cast(torch.randn(b, d, l, generator=G), dtype).requires_grad_(True)
        weight = torch.randn(d, w, generator=G).requires_grad_(True)            # fp32 weights: Vivim's case
        bias = torch.randn(d, generator=G).requires_grad_(True) if hb else None
        dout = cast(torch.randn(b, d, l, generator=G), dtype)
        out = cci.causal_conv1d_ref(x, weight, bias, "silu" if silu else None)
        out.backward(dout)
        save(name, x=npf(x), weight=npf(weight), bias=npf(bias), dout=npf(dout), out=npf(out),
             dx=npf(x.grad), dweight=npf(weight.grad), dbias=npf(bias.grad) if hb else None,
             meta=np.array([b, d, l, w, silu, hb]), dtype=np.array(dt))


def inner_params(G, d_inner, n, r, width=4):
    dtv = torch.exp(torch.rand(d_inner, generator=G) * (np.log(0.1) - np.log(1e-3)) + np.log(1e-3))
    return dict(
        conv_w=torch.randn(d_inner, 1, width, generator=G) * 0.3,
        conv_b=torch.randn(d_inner, generator=G) * 0.1,
        x_proj=torch.randn(r + 2 * n, d_inner, generator=G) * d_inner ** -0.5,
        dt_proj=(torch.rand(d_inner, r, generator=G) * 2 - 1) * r ** -0.5,
        A=-torch.arange(1, n + 1, dtype=torch.float32).repeat(d_inner, 1) * (0.5 + torch.rand(d_inner, n, generator=G)),
        D=torch.ones(d_inner) + 0.1 * torch.randn(d_inner, generator=G),
        dt_bias=dtv + torch.log(-torch.expm1(-dtv)),
    )


def make_inner(ssi):
    """Fused op: mamba_inner_ref (selective_scan_interface.py:636-670) with an identity out_proj,
    i.e. exactly MambaInnerFnNoOutProj's math (:155-225) transposed to (b, l, d)."""
    for i, (name, b, d_inner, n, r, l) in enumerate([("inner_small", 2, 16, 8, 2, 64),
                                                      ("inner_odd", 1, 8, 16, 1, 37)]):
        G = gen(300 + i)
        p = inner_params(G, d_inner, n, r)
        xz = torch.randn(b, 2 * d_inner, l, generator=G)
        dout = torch.randn(b, d_inner, l, generator=G)
        leaves = [xz] + list(p.values())
        for t in leaves:
            t.requires_grad_(True)
        eye = torch.eye(d_inner)
        y = ssi.mamba_inner_ref(xz, p["conv_w"], p["conv_b"], p["x_proj"], p["dt_proj"], eye, None,
                                p["A"], None, None, p["D"], delta_bias=p["dt_bias"], delta_softplus=True)
        out = y.transpose(1, 2)                      # back to (b, d, l) as NoOutProj returns
        out.backward(dout)
        save(name, xz=npf(xz), dout=npf(dout), out=npf(out), dxz=npf(xz.grad),
             **{k: npf(v) for k, v in p.items()}, **{"d" + k: npf(v.grad) for k, v in p.items()},
             meta=np.array([b, d_inner, n, r, l]))


def make_bimamba(ssi):
    """bimamba_inner_ref (selective_scan_interface.py:673-709): one conv / x_proj / dt_proj, a forward scan with A and a scan
    of the flipped sequence with A_b, summed, then out_proj.  Not on Vivim's v3 path (mamba_simple.py:125), but part of the
    surface `mamba_ssm` exports (bimamba_inner_fn, :616-625)."""
    for i, (name, b, d_inner, n, r, l, d_model) in enumerate([("bimamba_small", 2, 16, 8, 2, 64, 12),
                                                             ("bimamba_odd", 1, 8, 16, 1, 37, 5)]):
        G = gen(350 + i)
        p = inner_params(G, d_inner, n, r)
        p["A_b"] = -torch.arange(1, n + 1, dtype=torch.float32).repeat(d_inner, 1) * (0.5 + torch.rand(d_inner, n, generator=G))
        p["out_w"] = torch.randn(d_model, d_inner, generator=G) * d_inner ** -0.5
        p["out_b"] = torch.randn(d_model, generator=G) * 0.1
        xz = torch.randn(b, 2 * d_inner, l, generator=G)
        dout = torch.randn(b, l, d_model, generator=G)
        for t in [xz] + list(p.values()):
            t.requires_grad_(True)
        y = ssi.bimamba_inner_ref(xz, p["conv_w"], p["conv_b"], p["x_proj"], p["dt_proj"], p["out_w"], p["out_b"],
                                  p["A"], p["A_b"], None, None, p["D"], delta_bias=p["dt_bias"], delta_softplus=True)
        y.backward(dout)
        save(name, xz=npf(xz), dout=npf(dout), out=npf(y), dxz=npf(xz.grad),
             **{k: npf(v) for k, v in p.items()}, **{"d" + k: npf(v.grad) for k, v in p.items()},
             meta=np.array([b, d_inner, n, r, l, d_model]))


def make_module(cci, ssi, only=None):
    """v3 Mamba module forward/backward (mamba_simple.py:188-264).  The module's fast path calls
    mamba_inner_fn_no_out_proj (CUDA); it is rebound to the reference refs composed as in make_inner."""
    ms = _load("mamba_ssm.modules.mamba_simple", f"{REF}/mamba/mamba_ssm/modules/mamba_simple.py")

    def no_out_proj_via_refs(xz, conv_w, conv_b, x_proj, dt_proj, A, B=None, C=None, D=None,
                             delta_bias=None, B_proj_bias=None, C_proj_bias=None, delta_softplus=True):
        eye = torch.eye(A.shape[0], dtype=xz.dtype)
        y = ssi.mamba_inner_ref(xz, conv_w, conv_b, x_proj, dt_proj, eye, None, A, B, C, D,
                                delta_bias=delta_bias, delta_softplus=delta_softplus)
        return y.transpose(1, 2)

    ms.mamba_inner_fn_no_out_proj = no_out_proj_via_refs
    for i, (name, b, d_model, n, expand, nf, hw) in enumerate([("module_nf5", 2, 8, 4, 2, 5, 6),
                                                               ("module_nf3", 1, 8, 4, 2, 3, 4),
                                                               ("module_nf1", 1, 16, 8, 2, 1, 9),
                                                               # token counts that are multiples of 8: the build's
                                                               # grouped three-direction path (vivim_amd/mamba_simple.py)
                                                               ("module_nf5_hw8", 2, 16, 16, 2, 5, 8),
                                                               ("module_nf3_hw16", 1, 32, 16, 2, 3, 16),
                                                               # BASELINE.json configs[4]: d_state 64, expand 4, 8 frames
                                                               ("module_n64_e4_nf8", 1, 16, 64, 4, 8, 16)]):
        if only and name not in only:
            continue
        torch.manual_seed(400 + i)
        m = ms.Mamba(d_model=d_model, d_state=n, d_conv=4, expand=expand, bimamba_type="v3", nframes=nf)
        G = gen(450 + i)
        with torch.no_grad():           # de-symmetrise the three directions' A and D
            for k in ("A_log", "A_b_log", "A_s_log"):
                getattr(m, k).add_(0.2 * torch.randn(getattr(m, k).shape, generator=G))
            for k in ("D", "D_b", "D_s"):
                getattr(m, k).add_(0.2 * torch.randn(getattr(m, k).shape, generator=G))
        l = nf * hw
        x = torch.randn(b, l, d_model, generator=G).requires_grad_(True)
        dout = torch.randn(b, l, d_model, generator=G)
        y = m(x)
        y.backward(dout)
        sd = {"sd__" + k.replace(".", "__"): npf(v) for k, v in m.state_dict().items()}
        gr = {"grad__" + k.replace(".", "__"): npf(v.grad) for k, v in m.named_parameters()}
        save(name, x=npf(x), dout=npf(dout), y=npf(y), dx=npf(x.grad), **sd, **gr,
             meta=np.array([b, d_model, n, expand, nf, hw]))


def make_update(cci):
    """Single-token steps: causal_conv1d_update_ref (causal_conv1d_interface.py:83-104) and
    selective_state_update_ref (mamba_ssm/ops/triton/selective_state_update.py:157-192; the module imports triton at
    the top, which is installed here; only the pure-PyTorch ref is called)."""
    ssu = _load("ref_selective_state_update", f"{REF}/mamba/mamba_ssm/ops/triton/selective_state_update.py")
    for i, (name, b, d, w, hb, silu, dt) in enumerate([("update_conv_w4", 2, 96, 4, 1, 1, "fp32"),
                                                       ("update_conv_w2_plain", 3, 5, 2, 0, 0, "fp32"),
                                                       ("update_conv_w3_bf16", 2, 64, 3, 1, 1, "bf16")]):
        G = gen(600 + i)
        dtype = DT[dt]
        x = cast(torch.randn(b, d, generator=G), dtype)
        st = cast(torch.randn(b, d, w, generator=G), dtype)
        wt = torch.randn(d, w, generator=G)
        bias = torch.randn(d, generator=G) if hb else None
        st_new = st.clone()
        out = cci.causal_conv1d_update_ref(x, st_new, wt, bias, "silu" if silu else None)
        save(name, x=npf(x), conv_state=npf(st), weight=npf(wt), bias=npf(bias), out=npf(out), conv_state_new=npf(st_new),
             meta=np.array([b, d, w, hb, silu]), dtype=np.array(dt))
    for i, (name, b, d, n, hD, hZ, hb, sp, dt) in enumerate([("update_ssm_n16", 2, 64, 16, 1, 1, 1, 1, "fp32"),
                                                             ("update_ssm_plain", 1, 7, 8, 0, 0, 0, 0, "fp32"),
                                                             ("update_ssm_n64_bf16", 2, 32, 64, 1, 1, 1, 1, "bf16")]):
        G = gen(650 + i)
        dtype = DT[dt]
        state = torch.randn(b, d, n, generator=G)                       # fp32 state, as Mamba.step keeps it
        x = cast(torch.randn(b, d, generator=G), dtype)
        dtv = cast(0.5 * torch.rand(b, d, generator=G), dtype)
        A = -0.5 * torch.rand(d, n, generator=G) - 0.05
        Bm, Cm = cast(torch.randn(b, n, generator=G), dtype), cast(torch.randn(b, n, generator=G), dtype)
        D = torch.randn(d, generator=G) if hD else None
        z = cast(torch.randn(b, d, generator=G), dtype) if hZ else None
        bias = 0.5 * torch.rand(d, generator=G) if hb else None
        st_new = state.clone()
        out = ssu.selective_state_update_ref(st_new, x, dtv, A, Bm, Cm, D=D, z=z, dt_bias=bias, dt_softplus=bool(sp))
        save(name, state=npf(state), x=npf(x), dt=npf(dtv), A=npf(A), B=npf(Bm), C=npf(Cm), D=npf(D), z=npf(z),
             dt_bias=npf(bias), out=npf(out), state_new=npf(st_new), meta=np.array([b, d, n, hD, hZ, hb, sp]),
             dtype=np.array(dt))


if __name__ == "__main__":
    torch.set_num_threads(4)
    cci, ssi = load_reference()
    only = set(sys.argv[1:])          # e.g. `make_golden.py update module_nf5_hw8` regenerates just those
    if not only:
        make_scan(ssi)
        make_conv(cci)
        make_inner(ssi)
    if not only or "bimamba" in only:
        make_bimamba(ssi)
    if not only or "update" in only:
        make_update(cci)
    mods = {n for n in only if n.startswith("module_")}
    if not only or mods:
        make_module(cci, ssi, mods or None)
