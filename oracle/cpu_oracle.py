"""ctypes front-end of the C oracle (oracle/ssm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py -- never by vivim_amd/ or any product path.

Inputs are torch tensors of any float dtype / stride; they are up-cast to contiguous float32
(exact for fp16/bf16) before the C call, so low-precision parity tests compare the HIP result
with this fp64-accumulated result after rounding it to the I/O dtype.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libssm_oracle.so")


def build(force=False):
    """gcc-compile the oracle in place (seconds)."""
    src = os.path.join(_HERE, "ssm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f32(t):
    return None if t is None else t.detach().to("cpu", torch.float32).contiguous()


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def selective_scan_fwd(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False):
    """Returns (out, out_z_or_None, last_state) as float32 CPU tensors.
    Restates selective_scan_ref (selective_scan_interface.py:86-152)."""
    u, delta, A, B, C, D, z, delta_bias = map(_f32, (u, delta, A, B, C, D, z, delta_bias))
    batch, dim, L = u.shape
    N = A.shape[1]
    varB, varC = B.dim() >= 3, C.dim() >= 3
    if varB and B.dim() == 3:
        B = B.unsqueeze(1).contiguous()
    if varC and C.dim() == 3:
        C = C.unsqueeze(1).contiguous()
    G = B.shape[1] if varB else (C.shape[1] if varC else 1)
    out = torch.empty_like(u)
    out_z = torch.empty_like(u) if z is not None else None
    last = torch.empty(batch, dim, N)
    _load().oracle_selective_scan_fwd(
        _p(u), _p(delta), _p(A), _p(B), _p(C), _p(D), _p(z), _p(delta_bias), int(delta_softplus),
        int(varB), int(varC), batch, dim, L, N, G, _p(out), _p(out_z), _p(last))
    return out, out_z, last


def selective_scan_bwd(u, delta, A, B, C, D, z, delta_bias, dout, delta_softplus=False):
    """Returns dict(du, ddelta, dA, dB, dC, dD, dz, ddelta_bias) float32 CPU tensors
    (closed form of selective_scan_bwd_kernel.cuh:146-489)."""
    u, delta, A, B, C, D, z, delta_bias, dout = map(_f32, (u, delta, A, B, C, D, z, delta_bias, dout))
    batch, dim, L = u.shape
    N = A.shape[1]
    varB, varC = B.dim() >= 3, C.dim() >= 3
    sqB = varB and B.dim() == 3
    sqC = varC and C.dim() == 3
    if sqB:
        B = B.unsqueeze(1).contiguous()
    if sqC:
        C = C.unsqueeze(1).contiguous()
    G = B.shape[1] if varB else (C.shape[1] if varC else 1)
    du, ddelta = torch.empty_like(u), torch.empty_like(u)
    dA, dB, dC = torch.empty_like(A), torch.empty_like(B), torch.empty_like(C)
    dD = torch.empty(dim) if D is not None else None
    dz = torch.empty_like(u) if z is not None else None
    dbias = torch.empty(dim) if delta_bias is not None else None
    _load().oracle_selective_scan_bwd(
        _p(u), _p(delta), _p(A), _p(B), _p(C), _p(D), _p(z), _p(delta_bias), _p(dout),
        int(delta_softplus), int(varB), int(varC), batch, dim, L, N, G,
        _p(du), _p(ddelta), _p(dA), _p(dB), _p(dC), _p(dD), _p(dz), _p(dbias))
    if sqB:
        dB = dB.squeeze(1)
    if sqC:
        dC = dC.squeeze(1)
    return dict(du=du, ddelta=ddelta, dA=dA, dB=dB, dC=dC, dD=dD, dz=dz, ddelta_bias=dbias)


def causal_conv1d_fwd(x, weight, bias=None, silu=False):
    """causal_conv1d_ref (causal_conv1d_interface.py:49-65) -> float32 CPU tensor."""
    x, weight, bias = map(_f32, (x, weight, bias))
    batch, dim, L = x.shape
    out = torch.empty_like(x)
    _load().oracle_causal_conv1d_fwd(_p(x), _p(weight), _p(bias), int(silu), batch, dim, L,
                                     weight.shape[1], _p(out))
    return out


def causal_conv1d_bwd(x, weight, bias, dout, silu=False):
    """Closed form of causal_conv1d_bwd.cu:108-239 -> (dx, dweight, dbias_or_None)."""
    x, weight, bias, dout = map(_f32, (x, weight, bias, dout))
    batch, dim, L = x.shape
    dx = torch.empty_like(x)
    dw = torch.empty_like(weight)
    db = torch.empty(dim) if bias is not None else None
    _load().oracle_causal_conv1d_bwd(_p(x), _p(weight), _p(bias), _p(dout), int(silu), batch, dim, L,
                                     weight.shape[1], _p(dx), _p(dw), _p(db))
    return dx, dw, db


def causal_conv1d_update(x, conv_state, weight, bias=None, silu=False):
    """causal_conv1d_update_ref (causal_conv1d_interface.py:83-104) -> (out, new_conv_state) float32 CPU tensors
    (the input state is not modified)."""
    x, conv_state, weight, bias = map(_f32, (x, conv_state, weight, bias))
    conv_state = conv_state.clone()
    batch, dim = x.shape
    out = torch.empty_like(x)
    _load().oracle_causal_conv1d_update(_p(x), _p(conv_state), _p(weight), _p(bias), int(silu), batch, dim,
                                        weight.shape[1], _p(out))
    return out, conv_state


def selective_state_update(state, x, dt, A, B, C, D=None, z=None, dt_bias=None, dt_softplus=False):
    """selective_state_update_ref (selective_state_update.py:157-192) -> (out, new_state) float32 CPU tensors."""
    state, x, dt, A, B, C, D, z, dt_bias = map(_f32, (state, x, dt, A, B, C, D, z, dt_bias))
    state = state.clone()
    batch, dim, dstate = state.shape
    out = torch.empty_like(x)
    _load().oracle_selective_state_update(_p(state), _p(x), _p(dt), _p(A), _p(B), _p(C), _p(D), _p(z), _p(dt_bias),
                                          int(dt_softplus), batch, dim, dstate, _p(out))
    return out, state
