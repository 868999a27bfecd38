"""The v3 block's directional index maps as two HIP kernels (csrc/dirmap.hip; include/vivim_hip.h: vivim_dir_params).

    stack_directions(xz, nframes)      (B, 2D, L) -> (B, 2, 3, D, L) (channel-major in memory, see _scatter):
                                       [.., 0] = xz, [.., 1] = xz.flip(-1),
                                       [.., 2] = the frame interleave t*hw+p -> p*nf+t   (mamba_simple.py:231, 245-247)
    combine_directions(o3, nframes)    (B, 3, D, L) -> (B, D, L): (o3[:,0] + o3[:,1].flip(-1) + interleave^-1(o3[:,2])) / 3
                                       (mamba_simple.py:261-264)
Each reads its input once; each is the other's gradient (up to the scale)."""
import torch

from . import _lib

_ITYPE = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


def _params(flat, stacked, nframes, csplit, scale, src, dst):
    B, C, L = flat.shape
    P = _lib.DirParams()
    P.batch, P.channels, P.seqlen, P.nframes, P.csplit, P.itype, P.scale = B, C, L, nframes, csplit, _ITYPE[flat.dtype], scale
    P.flat_batch_stride, P.flat_c_stride = flat.stride(0), flat.stride(1)
    P.stk_batch_stride, P.stk_half_stride, P.stk_dir_stride, P.stk_c_stride = (stacked.stride(0), stacked.stride(1),
                                                                               stacked.stride(2), stacked.stride(3))
    P.src, P.dst = src.data_ptr(), dst.data_ptr()
    return P


def _scatter(flat, nframes, csplit, scale):
    """flat (B, C, L), unit L stride -> stacked (B, C / csplit, 3, csplit, L), CHANNEL-major in memory: the buffer is
    (C / csplit, 3, csplit, B, L) and the result its permuted view (batch stride L).  The fused grouped op multiplies
    all clips of a direction by that direction's weights in one GEMM over (channels) x (B * L) -- which needs the
    channel axis outermost -- and every kernel behind the C ABI takes batch / channel strides as they come."""
    if flat.stride(-1) != 1:
        flat = flat.contiguous()
    B, C, L = flat.shape
    stacked = _lib.empty((C // csplit, 3, csplit, B, L), flat.dtype, flat.device).permute(3, 0, 1, 2, 4)
    with torch.cuda.device(flat.device):
        _lib.call("vivim_dir_scatter", _params(flat, stacked, nframes, csplit, scale, flat, stacked),
                  torch.cuda.current_stream().cuda_stream)
    return stacked


def _gather(stacked, nframes, scale):
    """stacked (B, H, 3, csplit, L), unit L stride -> flat (B, H * csplit, L) contiguous."""
    if stacked.stride(-1) != 1:
        stacked = stacked.contiguous()
    B, H, _, csplit, L = stacked.shape
    flat = _lib.empty((B, H * csplit, L), stacked.dtype, stacked.device)
    with torch.cuda.device(stacked.device):
        _lib.call("vivim_dir_gather", _params(flat, stacked, nframes, csplit, scale, stacked, flat),
                  torch.cuda.current_stream().cuda_stream)
    return flat


class _StackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xz, nframes, halves):
        ctx.nframes = nframes
        return _scatter(xz, nframes, xz.shape[1] // halves, 1.0)

    @staticmethod
    def backward(ctx, g):
        return _gather(g, ctx.nframes, 1.0), None, None


class _CombineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, o3, nframes):
        ctx.nframes = nframes
        B, _, D, L = o3.shape
        return _gather(o3.view(B, 1, 3, D, L), nframes, 1.0 / 3.0)

    @staticmethod
    def backward(ctx, g):
        B, D, L = g.shape
        return _scatter(g, ctx.nframes, D, 1.0 / 3.0).view(B, 3, D, L), None


def stack_directions(xz, nframes, halves=2):
    return _StackFn.apply(xz, nframes, halves)


def combine_directions(o3, nframes):
    return _CombineFn.apply(o3, nframes)
