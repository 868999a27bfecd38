"""Autograd surface of the causal conv1d op: same names and behaviour as
causal-conv1d/causal_conv1d/causal_conv1d_interface.py:10-46 (CausalConv1dFn, causal_conv1d_fn)."""
import torch

from . import causal_conv1d_cuda

_ACTIVATIONS = (None, "silu", "swish")


class CausalConv1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias=None, activation=None):
        if activation not in _ACTIVATIONS:
            raise NotImplementedError("activation must be None, silu, or swish")
        if x.stride(2) != 1 and x.stride(1) != 1:
            x = x.contiguous()
        if bias is not None:
            bias = bias.contiguous()
        ctx.use_silu = activation is not None
        ctx.save_for_backward(x, weight, bias)
        return causal_conv1d_cuda.causal_conv1d_fwd(x, weight, bias, ctx.use_silu)

    @staticmethod
    def backward(ctx, dout):
        x, weight, bias = ctx.saved_tensors
        if dout.stride(2) != 1 and dout.stride(1) != 1:
            dout = dout.contiguous()
        dx, dweight, dbias = causal_conv1d_cuda.causal_conv1d_bwd(x, weight, bias, dout, None, ctx.use_silu)
        return dx, dweight, (dbias if bias is not None else None), None


def causal_conv1d_fn(x, weight, bias=None, activation=None):
    """x: (batch, dim, seqlen); weight: (dim, width); bias: (dim,); activation: None | "silu" | "swish"."""
    return CausalConv1dFn.apply(x, weight, bias, activation)


def causal_conv1d_update(x, conv_state, weight, bias=None, activation=None):
    """x (batch, dim), conv_state (batch, dim, width) advanced in place, weight (dim, width), bias (dim,)
    -> out (batch, dim)   (causal_conv1d_interface.py:68-80)."""
    if activation not in _ACTIVATIONS:
        raise NotImplementedError("activation must be None, silu, or swish")
    return causal_conv1d_cuda.causal_conv1d_update(x, conv_state, weight, bias, activation is not None)
