"""Drop-in for the reference's `causal_conv1d_cuda` extension module (pybind surface at
causal-conv1d/csrc/causal_conv1d.cpp:329-333): `causal_conv1d_fwd` / `causal_conv1d_bwd` with the same
positional signatures, checks and returns, on the gfx950 kernels behind include/vivim_hip.h.
`causal_conv1d_update` (single-token step for streaming inference) and the channel-last layout (unit stride along
channels) are built too; neither is on Vivim's training path.
"""
import torch

from . import _lib

_DT = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


def _check(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _checks(x, weight, bias_):
    """causal_conv1d.cpp:135-163."""
    _check(x.dtype in _DT, "causal_conv1d not implemented for input type '%s'" % x.dtype)
    _check(weight.dtype in _DT, "causal_conv1d not implemented for weight type '%s'" % weight.dtype)
    _check(x.is_cuda and weight.is_cuda, "x and weight must be CUDA/HIP tensors")
    _check(x.dim() == 3 and weight.dim() == 2, "x must be (batch, dim, seqlen) and weight (dim, width)")
    batch, dim, seqlen = x.shape
    width = weight.shape[-1]
    _check(tuple(weight.shape) == (dim, width), "weight must have shape (dim, width)")
    _check(x.stride(2) == 1 or x.stride(1) == 1, "x must have unit stride along seqlen or along channels")
    _check(2 <= width <= 4, "causal_conv1d only supports width between 2 and 4")
    if bias_ is not None:
        _check(bias_.dtype == weight.dtype and bias_.is_cuda and bias_.stride(-1) == 1
               and tuple(bias_.shape) == (dim,), "bias must be a contiguous (dim,) tensor of weight's dtype")
    return batch, dim, seqlen, width


def _fill(P, x, weight, bias_, silu_activation, dims):
    P.batch, P.dim, P.seqlen, P.width = dims
    P.itype, P.wtype = _DT[x.dtype], _DT[weight.dtype]
    P.silu_activation = int(bool(silu_activation))
    P.x_batch_stride, P.x_c_stride, P.x_l_stride = x.stride()
    P.weight_c_stride, P.weight_width_stride = weight.stride()
    P.x, P.weight = x.data_ptr(), weight.data_ptr()
    P.bias = None if bias_ is None else bias_.data_ptr()


def _channel_last(x):
    """(batch, dim, seqlen) with unit stride along channels and not along seqlen (causal_conv1d.cpp:151-152)."""
    return x.stride(1) == 1 and x.stride(2) > 1


def causal_conv1d_fwd(x, weight, bias_, silu_activation):
    """-> out (empty_like(x)); causal_conv1d.cpp:130-189."""
    dims = _checks(x, weight, bias_)
    if _channel_last(x):
        # lanes-along-channels kernels (csrc/conv1d_cl.hip; reference: causal_conv1d_fwd.cu:193-298); out keeps x's layout
        out = _lib.empty((x.shape[0], x.shape[2], x.shape[1]), x.dtype, x.device).transpose(1, 2)
    else:
        out = _lib.empty_like(x)
        if out.stride(2) != 1:       # empty_like of an exotic view may not keep unit seqlen stride
            out = _lib.empty(tuple(x.shape), x.dtype, x.device)
    P = _lib.ConvFwdParams()
    _fill(P, x, weight, bias_, silu_activation, dims)
    P.out = out.data_ptr()
    P.out_batch_stride, P.out_c_stride, P.out_l_stride = out.stride()
    with torch.cuda.device(x.device):
        _lib.call("vivim_causal_conv1d_fwd", P, torch.cuda.current_stream().cuda_stream)
    return out


def causal_conv1d_bwd(x, weight, bias_, dout, dx_, silu_activation):
    """-> [dx, dweight, dbias]; causal_conv1d.cpp:191-268."""
    dims = _checks(x, weight, bias_)
    batch, dim, seqlen, width = dims
    _check(dout.is_cuda and dout.dtype == x.dtype and tuple(dout.shape) == (batch, dim, seqlen),
           "dout must match x")
    if _channel_last(x):                              # csrc/conv1d_cl.hip (reference: causal_conv1d_bwd.cu:306-472)
        if dout.stride(1) != 1:
            dout = dout.transpose(-1, -2).contiguous().transpose(-1, -2)      # causal_conv1d.cpp:221
        if dx_ is not None:
            _check(dx_.dtype == x.dtype and dx_.is_cuda and tuple(dx_.shape) == (batch, dim, seqlen)
                   and dx_.stride(1) == 1, "dx must match x and have stride(1) == 1")       # causal_conv1d.cpp:237
            dx = dx_
        else:
            dx = _lib.empty((batch, seqlen, dim), x.dtype, x.device).transpose(1, 2)
    else:
        if dout.stride(2) != 1:
            dout = dout.contiguous()                  # causal_conv1d.cpp:220
        if dx_ is not None:
            _check(dx_.dtype == x.dtype and dx_.is_cuda and tuple(dx_.shape) == (batch, dim, seqlen)
                   and dx_.stride(2) == 1, "dx must match x and have stride(2) == 1")
            dx = dx_
        else:
            dx = _lib.empty_like(x)
            if dx.stride(2) != 1:
                dx = _lib.empty(tuple(x.shape), x.dtype, x.device)
    nw = dim * width                                  # one zero-filled fp32 accumulator for dweight | dbias: one fill
    acc = _lib.zeros(nw + (dim if bias_ is not None else 0), x.device)
    dweight = acc[:nw].view(dim, width)
    dbias = acc[nw:] if bias_ is not None else None
    P = _lib.ConvBwdParams()
    _fill(P.f, x, weight, bias_, silu_activation, dims)
    P.dout, P.dx, P.dweight = dout.data_ptr(), dx.data_ptr(), dweight.data_ptr()
    P.dbias = None if dbias is None else dbias.data_ptr()
    P.dout_batch_stride, P.dout_c_stride, P.dout_l_stride = dout.stride()
    P.dx_batch_stride, P.dx_c_stride, P.dx_l_stride = dx.stride()
    P.dweight_c_stride, P.dweight_width_stride = dweight.stride()
    with torch.cuda.device(x.device):
        _lib.call("vivim_causal_conv1d_bwd", P, torch.cuda.current_stream().cuda_stream)
    return [dx, dweight.to(weight.dtype), dbias.to(bias_.dtype) if bias_ is not None else None]


def causal_conv1d_update(x, conv_state, weight, bias_, silu_activation):
    """-> out (batch, dim); conv_state (batch, dim, width) is advanced in place (causal_conv1d.cpp:270-327)."""
    _check(x.dtype in _DT, "causal_conv1d_update not implemented for input type '%s'" % x.dtype)
    _check(weight.dtype in _DT, "causal_conv1d_update not implemented for weight type '%s'" % weight.dtype)
    _check(conv_state.dtype == x.dtype, "conv_state must have the dtype of x")
    _check(x.is_cuda and conv_state.is_cuda and weight.is_cuda, "x, conv_state and weight must be CUDA/HIP tensors")
    _check(x.dim() == 2 and weight.dim() == 2, "x must be (batch, dim) and weight (dim, width)")
    batch, dim = x.shape
    width = weight.shape[-1]
    _check(tuple(conv_state.shape) == (batch, dim, width), "conv_state must have shape (batch, dim, width)")
    _check(tuple(weight.shape) == (dim, width), "weight must have shape (dim, width)")
    _check(2 <= width <= 4, "causal_conv1d only supports width between 2 and 4")
    if bias_ is not None:
        _check(bias_.dtype == weight.dtype and bias_.is_cuda and bias_.stride(-1) == 1
               and tuple(bias_.shape) == (dim,), "bias must be a contiguous (dim,) tensor of weight's dtype")
    out = torch.empty_like(x)
    P = _lib.ConvUpdateParams()
    P.batch, P.dim, P.width = batch, dim, width
    P.itype, P.wtype, P.silu_activation = _DT[x.dtype], _DT[weight.dtype], int(bool(silu_activation))
    P.x_batch_stride, P.x_c_stride = x.stride()
    P.state_batch_stride, P.state_c_stride, P.state_w_stride = conv_state.stride()
    P.weight_c_stride, P.weight_width_stride = weight.stride()
    P.out_batch_stride, P.out_c_stride = out.stride()
    P.x, P.conv_state, P.weight, P.out = x.data_ptr(), conv_state.data_ptr(), weight.data_ptr(), out.data_ptr()
    P.bias = None if bias_ is None else bias_.data_ptr()
    with torch.cuda.device(x.device):
        _lib.call("vivim_causal_conv1d_update", P, torch.cuda.current_stream().cuda_stream)
    return out
