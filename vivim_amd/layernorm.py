"""LayerNorm over the channels of a channel-major token tensor (csrc/layernorm.hip; include/vivim_hip.h:
vivim_layernorm_params) -- the two norms around the Mamba call of MambaLayer (modeling/vivim.py:155-156).

    layer_norm_cm(x, weight, bias, eps)     x: (B, L, C) VIEW of (B, C, L) memory (token stride 1), as MambaLayer builds it
                                            -> (B, L, C) contiguous, the dtype F.layer_norm would return (f32 under autocast)
The transpose the ATen path does with a copy kernel in front of its row kernel is the kernel's own read pattern here.
`supported(x, weight)` says whether the fast path applies; the caller falls back to F.layer_norm otherwise (other layouts:
the ATen kernel is already the right one for token-major rows)."""
import ctypes

import torch

from . import _lib

_ITYPE = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


def supported(x, weight):
    if not (x.is_cuda and x.dim() == 3 and x.dtype in _ITYPE and weight is not None and weight.dtype == torch.float32):
        return False
    B, L, C = x.shape
    e = 16 // x.element_size()
    return (x.stride(1) == 1 and C <= 512 and L % e == 0 and x.stride(0) % e == 0 and x.stride(2) % e == 0
            and x.data_ptr() % 16 == 0 and x.stride(2) >= L)


def worthwhile(x):
    """Where the fused kernels beat the ATen path on the GPU AND repay the extra host time of a Python autograd node
    (profiles/r03_layernorm.log): from a few thousand tokens up -- stages 0 and 1 of the 256 x 256 configs (32 us against 184,
    28 against 75 per norm); at stages 2 and 3 (3 840 and 960 tokens) a tile per wave leaves the chip idle and ATen's row kernels
    are as fast or faster."""
    return x.shape[0] * x.shape[1] >= 4096


def _params(x, out_dtype, eps):
    B, L, C = x.shape
    P = _lib.LayerNormParams()
    P.batch, P.seqlen, P.channels, P.itype, P.otype, P.eps = B, L, C, _ITYPE[x.dtype], _ITYPE[out_dtype], eps
    P.x_batch_stride, P.x_c_stride = x.stride(0), x.stride(2)
    P.x = x.data_ptr()
    return P


def _launch(name, P, device):
    # the step is host-paced: no device context manager when the tensor's device is already the current one (~15 us each)
    if device.index == torch.cuda.current_device():
        _lib.call(name, P, torch.cuda.current_stream().cuda_stream)
    else:
        with torch.cuda.device(device):
            _lib.call(name, P, torch.cuda.current_stream().cuda_stream)


class _LayerNormCM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        B, L, C = x.shape
        y = _lib.empty((B, L, C), out_dtype, x.device)
        stats = _lib.empty((2, B, L), torch.float32, x.device)              # mean, rstd
        P = _params(x, out_dtype, eps)
        P.y_batch_stride, P.y_token_stride = L * C, C
        P.weight, P.bias = weight.data_ptr(), (bias.data_ptr() if bias is not None else None)
        P.y, P.mean = y.data_ptr(), stats.data_ptr()
        P.rstd = P.mean + 4 * B * L
        _launch("vivim_layernorm_cm_fwd", P, x.device)
        ctx.save_for_backward(x, weight, stats)
        ctx.eps, ctx.has_bias, ctx.out_dtype = eps, bias is not None, out_dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, stats = ctx.saved_tensors
        B, L, C = x.shape
        if dy.dtype != ctx.out_dtype:
            dy = dy.to(ctx.out_dtype)
        if dy.stride(2) != 1 or dy.stride(1) < C:
            dy = dy.contiguous()
        # dx in x's own layout: (B, C, L) memory seen as (B, L, C)
        dx = _lib.empty((B, C, L), x.dtype, x.device).transpose(1, 2)
        dwb = _lib.zeros(2 * C, x.device)                                    # dweight, dbias: one zero fill
        P = _params(x, ctx.out_dtype, ctx.eps)
        P.y_batch_stride, P.y_token_stride = dy.stride(0), dy.stride(1)
        P.dx_batch_stride, P.dx_c_stride = C * L, L
        P.weight, P.mean = weight.data_ptr(), stats.data_ptr()
        P.rstd = P.mean + 4 * B * L
        P.dy, P.dx, P.dweight = dy.data_ptr(), dx.data_ptr(), dwb.data_ptr()
        P.dbias = P.dweight + 4 * C if ctx.has_bias else None
        ws = _lib.empty((_lib.lib().vivim_layernorm_bwd_workspace_bytes(ctypes.byref(P)) // 4,), torch.float32, x.device)
        P.workspace = ws.data_ptr()                                          # per-tile dweight / dbias partial sums
        _launch("vivim_layernorm_cm_bwd", P, x.device)
        return dx, dwb[:C], (dwb[C:] if ctx.has_bias else None), None, None


def layer_norm_cm(x, weight, bias, eps=1e-5):
    """F.layer_norm(x, (C,), weight, bias, eps) for a channel-major x; output dtype as ATen's under the ambient autocast state."""
    out_dtype = torch.float32 if torch.is_autocast_enabled() else x.dtype
    return _LayerNormCM.apply(x, weight, bias, eps, out_dtype)
