"""Drop-in for the reference's `selective_scan_cuda` extension module (pybind surface at
mamba/csrc/selective_scan/selective_scan.cpp:494-497): `fwd` and `bwd` with the same positional
signatures, argument checks and return lists, running the gfx950 kernels through the C ABI
(include/vivim_hip.h).  Tensor allocation is the only thing PyTorch does here.
"""
import torch

from . import _lib

_ITYPE = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}
last_workspace_bytes = {}     # what the last fwd / bwd call asked for (tools/kbench.py derives the token-axis cut from it)


def _check(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _ptr(t):
    return None if t is None else t.data_ptr()


def chunk_len(P):
    """Tokens covered by one row of the checkpoint tensor `x` for the problem in `P` (a filled SsmFwdParams)."""
    return _lib.lib().vivim_scan_ckpt_len(P)


def _common_checks(u, delta, A, B, C, D_, z_, delta_bias_):
    """selective_scan.cpp:233-304 / 352-438."""
    _check(u.dtype in _ITYPE, "selective_scan not implemented for input type '%s'" % u.dtype)
    _check(A.dtype == torch.float32,
           "selective_scan not implemented for weight type '%s' (real fp32 A only)" % A.dtype)
    var_B, var_C = B.dim() >= 3, C.dim() >= 3
    _check(delta.dtype == u.dtype, "delta must have the dtype of u")
    _check(B.dtype == (u.dtype if var_B else A.dtype), "B has the wrong dtype")
    _check(C.dtype == (u.dtype if var_C else A.dtype), "C has the wrong dtype")
    for name, t in (("u", u), ("delta", delta), ("A", A), ("B", B), ("C", C)):
        _check(t.is_cuda, f"{name} must be a CUDA/HIP tensor")
    _check(u.dim() == 3, "u must be (batch, dim, seqlen)")
    _check(u.stride(-1) == 1 and delta.stride(-1) == 1, "u and delta must have stride(-1) == 1")
    batch, dim, seqlen = u.shape
    dstate = A.shape[1]
    n_groups = B.shape[1] if var_B else 1
    _check(dstate <= 256, "selective_scan only supports state dimension <= 256")
    _check(tuple(delta.shape) == (batch, dim, seqlen), "delta must have shape (batch, dim, seqlen)")
    _check(tuple(A.shape) == (dim, dstate), "A must have shape (dim, dstate)")
    if var_B:
        _check(B.dim() == 4 and tuple(B.shape) == (batch, n_groups, dstate, seqlen),
               "B must have shape (batch, n_groups, dstate, seqlen)")
        _check(B.stride(-1) == 1, "B must have stride(-1) == 1")
    else:
        _check(tuple(B.shape) == (dim, dstate), "B must have shape (dim, dstate)")
    if var_C:
        _check(C.dim() == 4 and tuple(C.shape) == (batch, n_groups, dstate, seqlen),
               "C must have shape (batch, n_groups, dstate, seqlen)")
        _check(C.stride(-1) == 1, "C must have stride(-1) == 1")
    else:
        _check(tuple(C.shape) == (dim, dstate), "C must have shape (dim, dstate)")
    for name, t in (("D", D_), ("delta_bias", delta_bias_)):
        if t is not None:
            _check(t.dtype == torch.float32 and t.is_cuda and t.stride(-1) == 1 and tuple(t.shape) == (dim,),
                   f"{name} must be a contiguous fp32 CUDA tensor of shape (dim,)")
    if z_ is not None:
        _check(z_.dtype == u.dtype and z_.is_cuda and z_.stride(-1) == 1
               and tuple(z_.shape) == (batch, dim, seqlen), "z must match u (dtype, shape, stride(-1) == 1)")
    return batch, dim, seqlen, dstate, n_groups, var_B, var_C


def _fill_fwd(P, u, delta, A, B, C, D_, z_, delta_bias_, delta_softplus, dims):
    batch, dim, seqlen, dstate, n_groups, var_B, var_C = dims
    P.batch, P.dim, P.seqlen, P.dstate, P.n_groups = batch, dim, seqlen, dstate, n_groups
    P.itype = _ITYPE[u.dtype]
    P.is_variable_B, P.is_variable_C = int(var_B), int(var_C)
    P.delta_softplus = int(bool(delta_softplus))
    P.u_batch_stride, P.u_d_stride = u.stride(0), u.stride(1)
    P.delta_batch_stride, P.delta_d_stride = delta.stride(0), delta.stride(1)
    P.A_d_stride, P.A_dstate_stride = A.stride(0), A.stride(1)
    if var_B:
        P.B_batch_stride, P.B_group_stride, P.B_dstate_stride = B.stride(0), B.stride(1), B.stride(2)
    else:
        P.B_batch_stride, P.B_group_stride, P.B_dstate_stride = 0, B.stride(0), B.stride(1)
    if var_C:
        P.C_batch_stride, P.C_group_stride, P.C_dstate_stride = C.stride(0), C.stride(1), C.stride(2)
    else:
        P.C_batch_stride, P.C_group_stride, P.C_dstate_stride = 0, C.stride(0), C.stride(1)
    P.u, P.delta, P.A, P.B, P.C = u.data_ptr(), delta.data_ptr(), A.data_ptr(), B.data_ptr(), C.data_ptr()
    P.D, P.delta_bias, P.z = _ptr(D_), _ptr(delta_bias_), _ptr(z_)
    if z_ is not None:
        P.z_batch_stride, P.z_d_stride = z_.stride(0), z_.stride(1)


def fwd(u, delta, A, B, C, D_, z_, delta_bias_, delta_softplus):
    """-> [out, x] (+ [out_z] when z is given); selective_scan.cpp:226-336."""
    dims = _common_checks(u, delta, A, B, C, D_, z_, delta_bias_)
    batch, dim, seqlen, dstate = dims[:4]
    out = _lib.empty_like(delta)                    # inherits delta's (L, B*L, 1) strides, selective_scan.cpp:311
    out_z = _lib.empty_like(z_) if z_ is not None else None
    P = _lib.SsmFwdParams()
    _fill_fwd(P, u, delta, A, B, C, D_, z_, delta_bias_, delta_softplus, dims)
    ck = chunk_len(P)
    x = _lib.empty((batch, dim, (seqlen + ck - 1) // ck, dstate), torch.float32, u.device)
    P.out, P.x = out.data_ptr(), x.data_ptr()
    P.out_batch_stride, P.out_d_stride = out.stride(0), out.stride(1)
    if out_z is not None:
        P.out_z = out_z.data_ptr()
        P.out_z_batch_stride, P.out_z_d_stride = out_z.stride(0), out_z.stride(1)
    ws_bytes = _lib.lib().vivim_scan_fwd_workspace_bytes(P)
    last_workspace_bytes["fwd"] = ws_bytes
    if ws_bytes:
        workspace = _lib.empty((ws_bytes,), torch.uint8, u.device)
        P.workspace, P.workspace_bytes = workspace.data_ptr(), ws_bytes
    with torch.cuda.device(u.device):
        _lib.call("vivim_selective_scan_fwd", P, torch.cuda.current_stream().cuda_stream)
    return [out, x] + ([out_z] if out_z is not None else [])


def bwd(u, delta, A, B, C, D_, z_, delta_bias_, dout, x_, out_, dz_, delta_softplus, recompute_out_z):
    """-> [du, ddelta, dA, dB, dC, dD, ddelta_bias] (+ [dz] if z) (+ [out_z] if recompute_out_z);
    selective_scan.cpp:338-492."""
    dims = _common_checks(u, delta, A, B, C, D_, z_, delta_bias_)
    batch, dim, seqlen, dstate, n_groups, var_B, var_C = dims
    _check(dout.dtype == u.dtype and dout.is_cuda and dout.stride(-1) == 1
           and tuple(dout.shape) == (batch, dim, seqlen), "dout must match u (dtype, shape, stride(-1) == 1)")
    P = _lib.SsmBwdParams()
    _fill_fwd(P.f, u, delta, A, B, C, D_, z_, delta_bias_, delta_softplus, dims)
    ck = chunk_len(P.f)
    n_chunks = (seqlen + ck - 1) // ck
    if n_chunks > 1:
        _check(x_ is not None, "x (scan checkpoints) is required when seqlen spans several chunks")
    if x_ is not None:
        _check(x_.dtype == torch.float32 and x_.is_cuda and x_.is_contiguous()
               and tuple(x_.shape) == (batch, dim, n_chunks, dstate),
               "x must be the contiguous fp32 (batch, dim, n_chunks, dstate) tensor returned by fwd")
    has_z = z_ is not None
    out_z = None
    dz = None
    if has_z:
        _check(out_ is not None, "out is required when z is given")
        _check(out_.dtype == u.dtype and out_.is_cuda and out_.stride(-1) == 1
               and tuple(out_.shape) == (batch, dim, seqlen), "out must match u (dtype, shape, stride(-1) == 1)")
        if dz_ is not None:
            _check(dz_.dtype == u.dtype and dz_.is_cuda and dz_.stride(-1) == 1
                   and tuple(dz_.shape) == (batch, dim, seqlen), "dz must match u (dtype, shape, stride(-1) == 1)")
            dz = dz_
        else:
            dz = _lib.empty_like(z_)
        if recompute_out_z:
            out_z = _lib.empty_like(out_)
    du = _lib.empty_like(u)
    ddelta = _lib.empty_like(delta)
    # fp32 accumulators (selective_scan.cpp:458-466) carved out of ONE zero-filled buffer: one memset
    # instead of five, and dB|dC are adjacent so their cast to the input dtype is one kernel too.
    nB, nC, nA = B.numel(), C.numel(), A.numel()
    acc = _lib.zeros(nB + nC + nA + 2 * dim, u.device)
    dB = acc[:nB].view(B.shape)
    dC = acc[nB:nB + nC].view(C.shape)
    dA = acc[nB + nC:nB + nC + nA].view(A.shape)
    dD = acc[nB + nC + nA:nB + nC + nA + dim] if D_ is not None else None
    ddelta_bias = acc[nB + nC + nA + dim:] if delta_bias_ is not None else None

    P.f.x = _ptr(x_)
    if has_z:
        P.f.out = out_.data_ptr()
        P.f.out_batch_stride, P.f.out_d_stride = out_.stride(0), out_.stride(1)
        P.dz = dz.data_ptr()
        P.dz_batch_stride, P.dz_d_stride = dz.stride(0), dz.stride(1)
        if out_z is not None:
            P.f.out_z = out_z.data_ptr()
            P.f.out_z_batch_stride, P.f.out_z_d_stride = out_z.stride(0), out_z.stride(1)
    P.dout = dout.data_ptr()
    P.dout_batch_stride, P.dout_d_stride = dout.stride(0), dout.stride(1)
    P.du, P.ddelta = du.data_ptr(), ddelta.data_ptr()
    P.du_batch_stride, P.du_d_stride = du.stride(0), du.stride(1)
    P.ddelta_batch_stride, P.ddelta_d_stride = ddelta.stride(0), ddelta.stride(1)
    P.dA, P.dB, P.dC = dA.data_ptr(), dB.data_ptr(), dC.data_ptr()
    P.dA_d_stride, P.dA_dstate_stride = dA.stride(0), dA.stride(1)
    if var_B:
        P.dB_batch_stride, P.dB_group_stride, P.dB_dstate_stride = dB.stride(0), dB.stride(1), dB.stride(2)
        P.dC_batch_stride, P.dC_group_stride, P.dC_dstate_stride = dC.stride(0), dC.stride(1), dC.stride(2)
    else:
        P.dB_batch_stride, P.dB_group_stride, P.dB_dstate_stride = 0, dB.stride(0), dB.stride(1)
        P.dC_batch_stride, P.dC_group_stride, P.dC_dstate_stride = 0, dC.stride(0), dC.stride(1)
    P.dD, P.ddelta_bias = _ptr(dD), _ptr(ddelta_bias)
    ws_bytes = _lib.lib().vivim_scan_bwd_workspace_bytes(P.f)
    last_workspace_bytes["bwd"] = ws_bytes
    if ws_bytes:
        workspace = _lib.empty((ws_bytes,), torch.uint8, u.device)   # torch caching allocator: no sync
        P.workspace, P.workspace_bytes = workspace.data_ptr(), ws_bytes
    with torch.cuda.device(u.device):
        _lib.call("vivim_selective_scan_bwd", P, torch.cuda.current_stream().cuda_stream)
    if var_B and var_C and B.dtype != torch.float32:
        dBC = acc[:nB + nC].to(B.dtype)
        dB_out, dC_out = dBC[:nB].view(B.shape), dBC[nB:].view(C.shape)
    else:
        dB_out, dC_out = dB.to(B.dtype), dC.to(C.dtype)
    result = [du, ddelta, dA, dB_out, dC_out, dD, ddelta_bias]
    if has_z:
        result.append(dz)
    if recompute_out_z:
        result.append(out_z)
    return result
