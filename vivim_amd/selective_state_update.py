"""`selective_state_update` -- the single-token SSM step of the reference
(mamba/mamba_ssm/ops/triton/selective_state_update.py:99-154, a Triton kernel there) on the gfx950 kernel behind
include/vivim_hip.h (csrc/update.hip).  Same signature, `state` is advanced in place."""
import torch

from . import _lib

_DT = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


def _check(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def selective_state_update(state, x, dt, A, B, C, D=None, z=None, dt_bias=None, dt_softplus=False):
    """state (batch, dim, dstate) [fp32 or x.dtype], x / dt (batch, dim), A (dim, dstate), B / C (batch, dstate),
    D (dim,), z (batch, dim), dt_bias (dim,)  ->  out (batch, dim)."""
    batch, dim, dstate = state.shape
    _check(x.dtype in _DT, "selective_state_update not implemented for input type '%s'" % x.dtype)
    _check(state.dtype in (torch.float32, x.dtype), "state must be fp32 or have the dtype of x")
    _check(tuple(x.shape) == (batch, dim) and tuple(dt.shape) == (batch, dim), "x and dt must be (batch, dim)")
    _check(tuple(A.shape) == (dim, dstate), "A must be (dim, dstate)")
    _check(tuple(B.shape) == (batch, dstate) and tuple(C.shape) == (batch, dstate), "B and C must be (batch, dstate)")
    _check(D is None or tuple(D.shape) == (dim,), "D must be (dim,)")
    _check(z is None or tuple(z.shape) == (batch, dim), "z must be (batch, dim)")
    _check(dt_bias is None or tuple(dt_bias.shape) == (dim,), "dt_bias must be (dim,)")
    for t in (state, x, dt, A, B, C):
        _check(t.is_cuda, "tensors must be CUDA/HIP tensors")
    cast = lambda t: None if t is None else t.to(x.dtype)
    dt, B, C, z = cast(dt), cast(B), cast(C), cast(z)
    f32 = lambda t: None if t is None else t.float().contiguous()
    A, D, dt_bias = A.float(), f32(D), f32(dt_bias)
    out = torch.empty_like(x)
    P = _lib.StateUpdateParams()
    P.batch, P.dim, P.dstate = batch, dim, dstate
    P.itype, P.stype, P.dt_softplus = _DT[x.dtype], _DT[state.dtype], int(bool(dt_softplus))
    P.state_batch_stride, P.state_d_stride, P.state_n_stride = state.stride()
    P.x_batch_stride, P.x_d_stride = x.stride()
    P.dt_batch_stride, P.dt_d_stride = dt.stride()
    P.A_d_stride, P.A_n_stride = A.stride()
    P.B_batch_stride, P.B_n_stride = B.stride()
    P.C_batch_stride, P.C_n_stride = C.stride()
    if z is not None:
        P.z_batch_stride, P.z_d_stride = z.stride()
    P.out_batch_stride, P.out_d_stride = out.stride()
    P.state, P.x, P.dt, P.A, P.B, P.C, P.out = (t.data_ptr() for t in (state, x, dt, A, B, C, out))
    P.D = None if D is None else D.data_ptr()
    P.z = None if z is None else z.data_ptr()
    P.dt_bias = None if dt_bias is None else dt_bias.data_ptr()
    with torch.cuda.device(x.device):
        _lib.call("vivim_selective_state_update", P, torch.cuda.current_stream().cuda_stream)
    return out
