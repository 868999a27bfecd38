"""Depthwise 3x3 / 3x3x3 convolution on token-major activations (B, D*H*W, C): the op behind
MambaLayer's Mlp (`DWConv`, modeling/vivim.py:57-68 -- nn.Conv3d(dim, dim, 3, 1, 1, groups=dim) applied to
x.transpose(1, 2).view(B, C, nf, H, W)) without the transposes, on the gfx950 kernels of csrc/dwconv.hip.
SURVEY.md section 8f row 4.  Same parameters as the nn.Conv3d / nn.Conv2d it replaces (weight (C,1,[kd,]3,3))."""
import os

import torch

from . import _lib

_DT = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


def supported(x, weight):
    """True when the kernels apply: CUDA tensor, channels contiguous, 3x3(x3) taps, aligned channel count."""
    if os.environ.get("VIVIM_NO_DWCONV") or not (x.is_cuda and x.dim() == 3 and x.dtype in _DT and x.stride(2) == 1):
        return False
    cv = 16 // x.element_size()
    ks = tuple(weight.shape[2:])
    return (weight.shape[1] == 1 and ks in ((3, 3), (3, 3, 3)) and x.shape[2] % cv == 0
            and x.stride(1) % cv == 0 and x.stride(0) % cv == 0 and x.data_ptr() % 16 == 0)


def _taps(weight):
    """(C, 1, [kd,] 3, 3) -> tap-major fp32 (kd*9, C)."""
    C = weight.shape[0]
    return weight.detach().reshape(C, -1).t().contiguous().float()


def _aligned(t):
    cv = 16 // t.element_size()
    if t.stride(2) != 1 or t.data_ptr() % 16 or t.stride(1) % cv or t.stride(0) % cv:
        t = t.contiguous()
    return t


def _run_fwd(x, wt, bias, dims, flip, act=0, aux=None):
    """act: 0 conv, 1 gelu(conv), 2 aux * gelu'(conv) (include/vivim_hip.h: vivim_dwconv_params.act)."""
    B, L, C = x.shape
    D, H, W = dims
    y = _lib.empty((B, L, C), x.dtype, x.device)
    P = _lib.DwConvParams()
    P.batch, P.depth, P.height, P.width, P.channels = B, D, H, W, C
    P.kd, P.itype, P.flip = wt.shape[0] // 9, _DT[x.dtype], int(flip)
    P.x_batch_stride, P.x_token_stride = x.stride(0), x.stride(1)
    P.y_batch_stride, P.y_token_stride = y.stride(0), y.stride(1)
    P.x, P.wt, P.y = x.data_ptr(), wt.data_ptr(), y.data_ptr()
    P.bias = None if bias is None else bias.data_ptr()
    P.act = act
    if aux is not None:
        P.aux, P.aux_batch_stride, P.aux_token_stride = aux.data_ptr(), aux.stride(0), aux.stride(1)
    with torch.cuda.device(x.device):
        _lib.call("vivim_dwconv_fwd", P, torch.cuda.current_stream().cuda_stream)
    return y


def _run_wgrad(x, dy, wt, dims, has_bias):
    """-> fp32 accumulator [taps * C | C]: dwt tap-major, dbias."""
    B, L, C = x.shape
    D, H, W = dims
    acc = _lib.zeros(wt.shape[0] * C + C, x.device)
    P = _lib.DwConvWgradParams()
    P.batch, P.depth, P.height, P.width, P.channels = B, D, H, W, C
    P.kd, P.itype = wt.shape[0] // 9, _DT[x.dtype]
    P.x_batch_stride, P.x_token_stride = x.stride(0), x.stride(1)
    P.dy_batch_stride, P.dy_token_stride = dy.stride(0), dy.stride(1)
    P.x, P.dy, P.dwt = x.data_ptr(), dy.data_ptr(), acc.data_ptr()
    P.dbias = acc.data_ptr() + 4 * wt.shape[0] * C if has_bias else None
    with torch.cuda.device(x.device):
        _lib.call("vivim_dwconv_wgrad", P, torch.cuda.current_stream().cuda_stream)
    return acc


class DepthwiseConvTokensFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, D, H, W):
        assert x.shape[1] == D * H * W, "token count must equal D*H*W"
        wt = _taps(weight)
        ctx.save_for_backward(x, wt)
        ctx.dims, ctx.wshape, ctx.wdtype = (D, H, W), weight.shape, weight.dtype
        ctx.has_bias = bias is not None
        ctx.bias_dtype = bias.dtype if bias is not None else None
        return _run_fwd(x, wt, bias.detach().float().contiguous() if bias is not None else None, (D, H, W), False)

    @staticmethod
    def backward(ctx, dy):
        x, wt = ctx.saved_tensors
        D, H, W = ctx.dims
        dy = _aligned(dy)
        dx = _run_fwd(dy, wt, None, (D, H, W), True) if ctx.needs_input_grad[0] else None
        C = x.shape[2]
        acc = _run_wgrad(x, dy, wt, (D, H, W), ctx.has_bias)
        dweight = acc[:wt.shape[0] * C].view(wt.shape[0], C).t().reshape(ctx.wshape).to(ctx.wdtype)
        dbias = acc[wt.shape[0] * C:].to(ctx.bias_dtype) if ctx.has_bias else None
        return dx, dweight, dbias, None, None, None


class DepthwiseConvGeluTokensFn(torch.autograd.Function):
    """gelu(dwconv(x)) as one node (the Mlp's `act(dwconv(fc1(x)))`, modeling/vivim.py:99-106):
    the activation is the convolution's epilogue, the pre-activation is never written -- the backward recomputes the
    convolution inside the kernel that multiplies the incoming gradient by gelu' (act = 2), then runs the same input- and
    weight-gradient kernels as the plain op.  erf-form GELU on the fp32 accumulator."""

    @staticmethod
    def forward(ctx, x, weight, bias, D, H, W):
        assert x.shape[1] == D * H * W, "token count must equal D*H*W"
        wt = _taps(weight)
        bf = bias.detach().float().contiguous() if bias is not None else None
        ctx.save_for_backward(x, wt, bf)
        ctx.dims, ctx.wshape, ctx.wdtype = (D, H, W), weight.shape, weight.dtype
        ctx.bias_dtype = bias.dtype if bias is not None else None
        return _run_fwd(x, wt, bf, (D, H, W), False, act=1)

    @staticmethod
    def backward(ctx, dy):
        x, wt, bf = ctx.saved_tensors
        dims = ctx.dims
        dpre = _run_fwd(x, wt, bf, dims, False, act=2, aux=_aligned(dy))        # dy * gelu'(conv(x) + bias)
        dx = _run_fwd(dpre, wt, None, dims, True) if ctx.needs_input_grad[0] else None
        C = x.shape[2]
        acc = _run_wgrad(x, dpre, wt, dims, bf is not None)
        dweight = acc[:wt.shape[0] * C].view(wt.shape[0], C).t().reshape(ctx.wshape).to(ctx.wdtype)
        dbias = acc[wt.shape[0] * C:].to(ctx.bias_dtype) if bf is not None else None
        return dx, dweight, dbias, None, None, None


def depthwise_conv_tokens(x, weight, bias, D, H, W):
    """x: (B, D*H*W, C) -> (B, D*H*W, C); weight (C, 1, 3, 3) with D == 1, or (C, 1, 3, 3, 3)."""
    return DepthwiseConvTokensFn.apply(x, weight, bias, D, H, W)


def depthwise_conv_gelu_tokens(x, weight, bias, D, H, W):
    """gelu(depthwise_conv_tokens(...)) with the activation fused into the convolution (erf form, nn.GELU's default)."""
    return DepthwiseConvGeluTokensFn.apply(x, weight, bias, D, H, W)
