"""Autograd surface of the selective scan and of the fused Mamba inner op.

Same names, argument order, saved-tensor policy and return tuples as the reference's
mamba/mamba_ssm/ops/selective_scan_interface.py:
  SelectiveScanFn / selective_scan_fn                       :14-83
  MambaInnerFnNoOutProj / mamba_inner_fn_no_out_proj        :155-289, :627-633   (what Vivim calls)
  BiMambaInnerFn / bimamba_inner_fn                         :437-603, :616-625
  mamba_inner_fn                                            :606-615  (composition with out_proj)
The CUDA extension calls are replaced by vivim_amd.selective_scan_cuda / causal_conv1d_cuda (gfx950
kernels behind the C ABI); the GEMMs inside the fused op stay on PyTorch-ROCm (hipBLASLt), as in the
reference where they are plain torch matmuls (:181-182, :272-277).
"""
import os

import torch
import torch.nn.functional as F
from torch.amp import custom_bwd, custom_fwd

from . import causal_conv1d_cuda, selective_scan_cuda
from . import wgrad as _wg


def _unit_l(t):
    return t if t is None or t.stride(-1) == 1 else t.contiguous()


class SelectiveScanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                return_last_state=False):
        u, delta, B, C, z = (_unit_l(t) for t in (u, delta, B, C, z))
        if D is not None:
            D = D.contiguous()
        ctx.squeeze_B = B.dim() == 3
        ctx.squeeze_C = C.dim() == 3
        if ctx.squeeze_B:
            B = B.unsqueeze(1)
        if ctx.squeeze_C:
            C = C.unsqueeze(1)
        out, x, *rest = selective_scan_cuda.fwd(u, delta, A, B, C, D, z, delta_bias, delta_softplus)
        ctx.delta_softplus = delta_softplus
        ctx.has_z = z is not None
        last_state = x[:, :, -1, :]                       # (batch, dim, dstate); reference: x[:, :, -1, 1::2]
        if ctx.has_z:
            ctx.save_for_backward(u, delta, A, B, C, D, z, delta_bias, x, out)
            result = rest[0]
        else:
            ctx.save_for_backward(u, delta, A, B, C, D, delta_bias, x)
            result = out
        return (result, last_state) if return_last_state else result

    @staticmethod
    def backward(ctx, dout, *ignored):
        if ctx.has_z:
            u, delta, A, B, C, D, z, delta_bias, x, out = ctx.saved_tensors
        else:
            u, delta, A, B, C, D, delta_bias, x = ctx.saved_tensors
            z = out = None
        dout = _unit_l(dout)
        du, ddelta, dA, dB, dC, dD, ddelta_bias, *rest = selective_scan_cuda.bwd(
            u, delta, A, B, C, D, z, delta_bias, dout, x, out, None, ctx.delta_softplus, False)
        dz = rest[0] if ctx.has_z else None
        if ctx.squeeze_B:
            dB = dB.squeeze(1)
        if ctx.squeeze_C:
            dC = dC.squeeze(1)
        return (du, ddelta, dA, dB, dC, dD if D is not None else None, dz,
                ddelta_bias if delta_bias is not None else None, None, None)


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False):
    """If return_last_state, returns (out, last_state) with last_state (batch, dim, dstate); the gradient
    of last_state is not propagated (as in the reference, :79-82)."""
    return SelectiveScanFn.apply(u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state)


class MambaInnerFnNoOutProj(torch.autograd.Function):
    """conv1d+SiLU -> x_proj -> dt_proj -> selective scan (+D, gated by silu(z)) as ONE autograd node with
    activation recompute: conv1d_out and delta are dropped after the forward and rebuilt in the backward
    (checkpoint_lvl=1, reference :218-219, :238-241)."""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None,
                delta_softplus=True, checkpoint_lvl=1):
        assert checkpoint_lvl in (0, 1)
        if A.is_complex():
            raise NotImplementedError("complex A is outside Vivim's path (A is real fp32, mamba_simple.py:212)")
        batch, _, L = xz.shape
        R = delta_proj_weight.shape[1]
        N = A.shape[-1]
        if torch.is_autocast_enabled("cuda"):
            amp_dtype = torch.get_autocast_dtype("cuda")
            x_proj_weight = x_proj_weight.to(amp_dtype)
            delta_proj_weight = delta_proj_weight.to(amp_dtype)
        xz = _unit_l(xz)
        conv1d_weight = conv1d_weight.squeeze(1)                       # (d, 1, w) -> (d, w)
        x, z = xz.chunk(2, dim=1)
        conv1d_bias = conv1d_bias.contiguous() if conv1d_bias is not None else None
        conv1d_out = causal_conv1d_cuda.causal_conv1d_fwd(x, conv1d_weight, conv1d_bias, True)
        d_inner = conv1d_out.shape[1]
        # (b d l) -> ((b l) d) @ W_x^T : (b l, R + 2N); delta keeps d slowest / l fastest for the scan
        x_dbl = F.linear(conv1d_out.transpose(1, 2).reshape(batch * L, d_inner), x_proj_weight)
        delta = (delta_proj_weight @ x_dbl[:, :R].t()).view(d_inner, batch, L).transpose(0, 1)
        ctx.is_variable_B = B is None
        ctx.is_variable_C = C is None
        ctx.B_proj_bias_is_None = B_proj_bias is None
        ctx.C_proj_bias_is_None = C_proj_bias is None
        if B is None:
            B = x_dbl[:, R:R + N]
            if B_proj_bias is not None:
                B = B + B_proj_bias.to(B.dtype)
            B = B.view(batch, L, N).transpose(1, 2).unsqueeze(1).contiguous()   # (b, 1, N, l)
        else:
            B = _unit_l(B)
        if C is None:
            C = x_dbl[:, -N:]
            if C_proj_bias is not None:
                C = C + C_proj_bias.to(C.dtype)
            C = C.view(batch, L, N).transpose(1, 2).unsqueeze(1).contiguous()
        else:
            C = _unit_l(C)
        if D is not None:
            D = D.contiguous()
        out, scan_intermediates, out_z = selective_scan_cuda.fwd(
            conv1d_out, delta, A, B, C, D, z, delta_bias, delta_softplus)
        ctx.delta_softplus = delta_softplus
        ctx.checkpoint_lvl = checkpoint_lvl
        if checkpoint_lvl >= 1:
            conv1d_out = delta = None
        ctx.save_for_backward(xz, conv1d_weight, conv1d_bias, x_dbl, x_proj_weight, delta_proj_weight,
                              conv1d_out, delta, A, B, C, D, delta_bias, scan_intermediates, out)
        return out_z

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dout):
        (xz, conv1d_weight, conv1d_bias, x_dbl, x_proj_weight, delta_proj_weight, conv1d_out, delta,
         A, B, C, D, delta_bias, scan_intermediates, out) = ctx.saved_tensors
        batch, _, L = xz.shape
        R = delta_proj_weight.shape[1]
        N = A.shape[-1]
        x, z = xz.chunk(2, dim=1)
        d_inner = x.shape[1]
        dout = _unit_l(dout)
        if ctx.checkpoint_lvl == 1:
            conv1d_out = causal_conv1d_cuda.causal_conv1d_fwd(x, conv1d_weight, conv1d_bias, True)
            delta = (delta_proj_weight @ x_dbl[:, :R].t()).view(d_inner, batch, L).transpose(0, 1)
        # dx and dz are written straight into the two halves of dxz (no torch.cat), reference :244-251
        dxz = torch.empty_like(xz)
        dx, dz = dxz.chunk(2, dim=1)
        # The reference asks the kernel to recompute out_z here and drops it (:247-251); we do not ask.
        dconv1d_out, ddelta, dA, dB, dC, dD, ddelta_bias, dz = selective_scan_cuda.bwd(
            conv1d_out, delta, A, B, C, D, z, delta_bias, dout, scan_intermediates, out, dz,
            ctx.delta_softplus, False)
        dx_dbl = torch.empty_like(x_dbl)
        dB_proj_bias = dC_proj_bias = None
        # With a constant B (or C) the projection's B (C) columns feed nothing: their gradient is zero.  (The reference
        # leaves them uninitialised, :262-271 -- harmless only while the allocator hands out zeroed memory.)
        if ctx.is_variable_B:
            dB = dB.squeeze(1).transpose(1, 2).reshape(batch * L, N)       # (b 1 N l) -> ((b l) N)
            dB_proj_bias = dB.sum(0) if not ctx.B_proj_bias_is_None else None
            dx_dbl[:, R:R + N] = dB
            dB = None
        else:
            dx_dbl[:, R:R + N].zero_()
        if ctx.is_variable_C:
            dC = dC.squeeze(1).transpose(1, 2).reshape(batch * L, N)
            dC_proj_bias = dC.sum(0) if not ctx.C_proj_bias_is_None else None
            dx_dbl[:, -N:] = dC
            dC = None
        else:
            dx_dbl[:, -N:].zero_()
        ddelta = ddelta.transpose(0, 1).reshape(d_inner, batch * L)        # (b d l) -> (d (b l))
        ddelta_proj_weight = ddelta @ x_dbl[:, :R]
        dx_dbl[:, :R] = ddelta.t() @ delta_proj_weight
        dconv1d_out = dconv1d_out.transpose(0, 1).reshape(d_inner, batch * L)
        dx_proj_weight = dx_dbl.t() @ conv1d_out.transpose(1, 2).reshape(batch * L, d_inner)
        dconv1d_out = torch.addmm(dconv1d_out, x_proj_weight.t(), dx_dbl.t())
        dconv1d_out = dconv1d_out.view(d_inner, batch, L).transpose(0, 1)  # -> (b d l), unit l stride
        dx, dconv1d_weight, dconv1d_bias = causal_conv1d_cuda.causal_conv1d_bwd(
            x, conv1d_weight, conv1d_bias, dconv1d_out, dx, True)
        return (dxz, dconv1d_weight.unsqueeze(1), dconv1d_bias if conv1d_bias is not None else None,
                dx_proj_weight, ddelta_proj_weight, dA, dB, dC, dD if D is not None else None,
                ddelta_bias if delta_bias is not None else None, dB_proj_bias, dC_proj_bias, None, None)


def mamba_inner_fn_no_out_proj(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                               A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
                               C_proj_bias=None, delta_softplus=True):
    """xz: (batch, 2*d_inner, seqlen) -> (batch, d_inner, seqlen)."""
    return MambaInnerFnNoOutProj.apply(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                                       A, B, C, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus)


def _channel_major(t):
    """(batch, C, l) -> the same values laid out (C, batch, l) in memory, returned as the (batch, C, l) view with strides
    (l, batch*l, 1).  A no-op for tensors that already are (Vivim's xz, everything the grouped op allocates)."""
    b, c, l = t.shape
    if t.stride() == (l, b * l, 1) or b == 1 and t.stride(2) == 1 and t.stride(1) == l:
        return t
    return t.permute(1, 0, 2).contiguous().permute(1, 0, 2)


class MambaInnerGroupedFnNoOutProj(torch.autograd.Function):
    """G independent `MambaInnerFnNoOutProj` problems (Vivim: the three scan directions of one v3 block, each with
    its own conv / x_proj / dt_proj / A / D) as ONE autograd node: the directions are laid side by side on the
    channel axis, so there is one conv1d launch over G*D channels and one scan launch with n_groups = G instead of
    G of each, and G times more independent work per launch.

    Everything inside is CHANNEL-major: conv_out, delta, x_dbl live as (G, D | R+2N, batch*l), i.e. the (batch, C, l) views
    the kernels see have strides (l, batch*l, 1) -- the layout of Vivim's own xz (mamba_simple.py:204-208).  Each
    projection is then ONE batched GEMM over the G directions with all clips in its N dimension,
    x_dbl = W_x @ conv_out: no per-clip weight replication, weight gradients summed over the clips inside the GEMM's fp32
    accumulator, and B / C are (b, G, N, l) VIEWS of x_dbl with unit l-stride -- the `(b l) N -> b 1 N l` transpose copies
    of the single-direction op (reference :193, :205) and the transposed copy of conv_out for `F.linear` (:181) are gone.
    Same math, same saved-tensor policy (conv_out and delta are rebuilt in the backward, checkpoint_lvl = 1).

    xz:   (batch, 2, G, D, l), unit l stride -- [:, 0] the x halves, [:, 1] the z halves of the G directions; channel-major
          as `dirmap.stack_directions` makes it (any other layout is re-laid once)
    conv1d_weight (G*D, W), conv1d_bias (G*D) | x_proj_weight (G, R + 2N, D) | delta_proj_weight (G, D, R)
    A (G*D, N) fp32 | D (G*D) fp32 | delta_bias (G*D) fp32            ->  out_z (batch, G*D, l)
    """

    @staticmethod
    def _cm(t, G):
        """(batch, G*D, l) with strides (l, batch*l, 1) -> (G, D, batch*l) view."""
        b, c, l = t.shape
        return t.permute(1, 0, 2).reshape(G, c // G, b * l)

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, D, delta_bias,
                delta_softplus=True):
        batch, two, G, Dm, L = xz.shape
        assert two == 2
        R = delta_proj_weight.shape[2]
        N = A.shape[-1]
        if torch.is_autocast_enabled("cuda"):
            amp_dtype = torch.get_autocast_dtype("cuda")
            x_proj_weight = x_proj_weight.to(amp_dtype)
            delta_proj_weight = delta_proj_weight.to(amp_dtype)
        if xz.stride() != (L, G * Dm * batch * L, Dm * batch * L, batch * L, 1):
            xz = xz.permute(1, 2, 3, 0, 4).contiguous().permute(3, 0, 1, 2, 4)
        x = xz[:, 0].reshape(batch, G * Dm, L)            # views with strides (l, batch*l, 1)
        z = xz[:, 1].reshape(batch, G * Dm, L)
        conv1d_bias = conv1d_bias.contiguous() if conv1d_bias is not None else None
        conv1d_out = _channel_major(causal_conv1d_cuda.causal_conv1d_fwd(x, conv1d_weight, conv1d_bias, True))
        cm = MambaInnerGroupedFnNoOutProj._cm
        x_dbl = torch.bmm(x_proj_weight, cm(conv1d_out, G))                       # (G, R + 2N, batch*l)
        delta = torch.bmm(delta_proj_weight, x_dbl[:, :R]).view(G * Dm, batch, L).permute(1, 0, 2)
        x_dbl4 = x_dbl.view(G, -1, batch, L)
        B, C = x_dbl4[:, R:R + N].permute(2, 0, 1, 3), x_dbl4[:, R + N:].permute(2, 0, 1, 3)   # (batch, G, N, l) views
        out, scan_intermediates, out_z = selective_scan_cuda.fwd(
            conv1d_out, delta, A, B, C, D.contiguous(), z, delta_bias, delta_softplus)
        ctx.delta_softplus = delta_softplus
        ctx.save_for_backward(xz, conv1d_weight, conv1d_bias, x_dbl, x_proj_weight, delta_proj_weight,
                              A, D, delta_bias, scan_intermediates, out)
        return out_z

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dout):
        (xz, conv1d_weight, conv1d_bias, x_dbl, x_proj_weight, delta_proj_weight, A, D, delta_bias,
         scan_intermediates, out) = ctx.saved_tensors
        batch, _, G, Dm, L = xz.shape
        E, R = x_proj_weight.shape[1], delta_proj_weight.shape[2]
        N = A.shape[-1]
        cm = MambaInnerGroupedFnNoOutProj._cm
        x = xz[:, 0].reshape(batch, G * Dm, L)
        z = xz[:, 1].reshape(batch, G * Dm, L)
        dout = _unit_l(dout)
        conv1d_out = _channel_major(causal_conv1d_cuda.causal_conv1d_fwd(x, conv1d_weight, conv1d_bias, True))
        conv_g = cm(conv1d_out, G)                                                # (G, D, batch*l)
        delta = torch.bmm(delta_proj_weight, x_dbl[:, :R]).view(G * Dm, batch, L).permute(1, 0, 2)
        x_dbl4 = x_dbl.view(G, E, batch, L)
        B, C = x_dbl4[:, R:R + N].permute(2, 0, 1, 3), x_dbl4[:, R + N:].permute(2, 0, 1, 3)
        dxz = torch.empty_like(xz)                                                # keeps xz's channel-major strides
        dx = dxz[:, 0].reshape(batch, G * Dm, L)
        dz = dxz[:, 1].reshape(batch, G * Dm, L)
        dconv1d_out, ddelta, dA, dB, dC, dD, ddelta_bias, dz = selective_scan_cuda.bwd(
            conv1d_out, delta, A, B, C, D, z, delta_bias, dout, scan_intermediates, out, dz,
            ctx.delta_softplus, False)
        ddelta_g = cm(_channel_major(ddelta), G)                                  # (G, D, batch*l)
        dx_dbl = torch.empty_like(x_dbl)                                          # (G, E, batch*l)
        dx_dbl4 = dx_dbl.view(G, E, batch, L)
        dx_dbl4[:, R:R + N] = dB.permute(1, 2, 0, 3)
        dx_dbl4[:, R + N:] = dC.permute(1, 2, 0, 3)
        dx_dbl[:, :R] = torch.bmm(delta_proj_weight.transpose(1, 2), ddelta_g)
        # weight gradients: one GEMM per direction over all clips and tokens (fp32 accumulation inside the GEMM)
        # (K = every token of every clip against 4 x 128 or 36 x 128 outputs: the library GEMM puts such a product on two to
        # four workgroups, 77-100 us each at stage 0; csrc/wgrad.hip splits the token axis instead)
        x_r = x_dbl[:, :R]
        if _wg.supported(ddelta_g, x_r) and _wg.supported(dx_dbl, conv_g) and not os.environ.get("VIVIM_NO_WGRAD_KERNEL"):
            ddelta_proj_weight = _wg.wgrad_nt(ddelta_g, x_r).to(delta_proj_weight.dtype)        # (G, D, R)
            dx_proj_weight = _wg.wgrad_nt(dx_dbl, conv_g).to(x_proj_weight.dtype)               # (G, E, D)
        else:
            ddelta_proj_weight = torch.bmm(ddelta_g, x_r.transpose(1, 2))
            dx_proj_weight = torch.bmm(dx_dbl, conv_g.transpose(1, 2))
        dconv_g = torch.baddbmm(cm(_channel_major(dconv1d_out), G), x_proj_weight.transpose(1, 2), dx_dbl)
        dconv = dconv_g.view(G * Dm, batch, L).permute(1, 0, 2)
        dx, dconv1d_weight, dconv1d_bias = causal_conv1d_cuda.causal_conv1d_bwd(
            x, conv1d_weight, conv1d_bias, dconv, dx, True)
        return (dxz, dconv1d_weight, dconv1d_bias if conv1d_bias is not None else None, dx_proj_weight,
                ddelta_proj_weight, dA, dD, ddelta_bias, None)


def mamba_inner_grouped_fn_no_out_proj(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, D,
                                       delta_bias, delta_softplus=True):
    return MambaInnerGroupedFnNoOutProj.apply(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                                              A, D, delta_bias, delta_softplus)


def mamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                   out_proj_weight, out_proj_bias, A, B=None, C=None, D=None, delta_bias=None,
                   B_proj_bias=None, C_proj_bias=None, delta_softplus=True):
    """Same math as the reference's MambaInnerFn (:292-434): the fused inner op followed by out_proj."""
    y = mamba_inner_fn_no_out_proj(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                                   A, B, C, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus)
    return F.linear(y.transpose(1, 2), out_proj_weight, out_proj_bias)


class BiMambaInnerFn(torch.autograd.Function):
    """The reference's BiMambaInnerFn (:437-603): one conv / x_proj / dt_proj, the scan of the sequence with A plus the scan
    of the time-reversed sequence with A_b (reversed back), out_proj; same arguments, same saved-tensor policy
    (checkpoint_lvl = 1: conv1d_out and delta are rebuilt in the backward), same gradient tuple.  Dead code in Vivim's fork
    (mamba_simple.py:125 asserts v3) but part of what `mamba_ssm` exports (:616-625).

    Where the reference launches the scan twice per pass (:499-505, :541-553), the two directions here sit side by side on
    the channel axis -- channels [0, d) the sequence, [d, 2d) its reversal, B / C as two groups -- so each pass is ONE
    forward and ONE backward scan launch with n_groups = 2 and twice the independent work per launch."""

    @staticmethod
    def _both(t, dim=1):
        """(b, c, l) -> (b, 2c, l): the tensor and its time reversal side by side on `dim`."""
        return torch.cat([t, t.flip([-1])], dim=dim)

    @staticmethod
    def _fold(t2):
        """(b, 2c, l) -> (b, c, l): first half + reversed second half."""
        c = t2.shape[1] // 2
        return t2[:, :c] + t2[:, c:].flip([-1])

    @staticmethod
    def _stack(conv1d_out, delta, z, B, C, A, A_b, D, delta_bias, var_B, var_C):
        both = BiMambaInnerFn._both
        twice = lambda t: None if t is None else torch.cat([t, t])
        return (both(conv1d_out), both(delta), both(z), both(B) if var_B else twice(B), both(C) if var_C else twice(C),
                torch.cat([A, A_b]), twice(D), twice(delta_bias))

    @staticmethod
    def _project(x_dbl, delta_proj_weight, B, C, B_proj_bias, C_proj_bias, batch, L, N):
        R = delta_proj_weight.shape[1]
        d_inner = delta_proj_weight.shape[0]
        delta = (delta_proj_weight @ x_dbl[:, :R].t()).view(d_inner, batch, L).transpose(0, 1)
        if B is None:
            B = x_dbl[:, R:R + N]
            if B_proj_bias is not None:
                B = B + B_proj_bias.to(B.dtype)
            B = B.view(batch, L, N).transpose(1, 2).unsqueeze(1)           # (b, 1, N, l) view; `_both` copies it
        if C is None:
            C = x_dbl[:, -N:]
            if C_proj_bias is not None:
                C = C + C_proj_bias.to(C.dtype)
            C = C.view(batch, L, N).transpose(1, 2).unsqueeze(1)
        return delta, B, C

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias,
                A, A_b, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None,
                delta_softplus=True, checkpoint_lvl=1):
        assert checkpoint_lvl in (0, 1)
        if A.is_complex() or A_b.is_complex():
            raise NotImplementedError("complex A is outside this build (the reference asserts a real A_b, :503)")
        batch, _, L = xz.shape
        N = A.shape[-1]
        if torch.is_autocast_enabled("cuda"):
            amp_dtype = torch.get_autocast_dtype("cuda")
            x_proj_weight = x_proj_weight.to(amp_dtype)
            delta_proj_weight = delta_proj_weight.to(amp_dtype)
            out_proj_weight = out_proj_weight.to(amp_dtype)
            out_proj_bias = out_proj_bias.to(amp_dtype) if out_proj_bias is not None else None
        xz = _unit_l(xz)
        conv1d_weight = conv1d_weight.squeeze(1)
        x, z = xz.chunk(2, dim=1)
        conv1d_bias = conv1d_bias.contiguous() if conv1d_bias is not None else None
        conv1d_out = causal_conv1d_cuda.causal_conv1d_fwd(x, conv1d_weight, conv1d_bias, True)
        d_inner = conv1d_out.shape[1]
        x_dbl = F.linear(conv1d_out.transpose(1, 2).reshape(batch * L, d_inner), x_proj_weight)
        ctx.is_variable_B, ctx.is_variable_C = B is None, C is None
        ctx.B_proj_bias, ctx.C_proj_bias = B_proj_bias, C_proj_bias
        delta, Bv, Cv = BiMambaInnerFn._project(x_dbl, delta_proj_weight, B, C, B_proj_bias, C_proj_bias, batch, L, N)
        u2, delta2, z2, B2, C2, A2, D2, bias2 = BiMambaInnerFn._stack(
            conv1d_out, delta, z, Bv, Cv, A, A_b, D, delta_bias, ctx.is_variable_B, ctx.is_variable_C)
        out2, scan_intermediates, out_z2 = selective_scan_cuda.fwd(u2, delta2, A2, B2, C2, D2, z2, bias2, delta_softplus)
        out_z = BiMambaInnerFn._fold(out_z2)
        ctx.delta_softplus = delta_softplus
        ctx.out_proj_bias_is_None = out_proj_bias is None
        ctx.checkpoint_lvl = checkpoint_lvl
        if checkpoint_lvl >= 1:
            conv1d_out = delta = None
        ctx.save_for_backward(xz, conv1d_weight, conv1d_bias, x_dbl, x_proj_weight, delta_proj_weight, out_proj_weight,
                              conv1d_out, delta, A, A_b, B, C, D, delta_bias, scan_intermediates, out2)
        return F.linear(out_z.transpose(1, 2), out_proj_weight, out_proj_bias)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dout):
        (xz, conv1d_weight, conv1d_bias, x_dbl, x_proj_weight, delta_proj_weight, out_proj_weight, conv1d_out, delta,
         A, A_b, B, C, D, delta_bias, scan_intermediates, out2) = ctx.saved_tensors
        batch, _, L = xz.shape
        R = delta_proj_weight.shape[1]
        N = A.shape[-1]
        x, z = xz.chunk(2, dim=1)
        d_inner = x.shape[1]
        if ctx.checkpoint_lvl == 1:
            conv1d_out = causal_conv1d_cuda.causal_conv1d_fwd(x, conv1d_weight, conv1d_bias, True)
        delta_r, Bv, Cv = BiMambaInnerFn._project(x_dbl, delta_proj_weight, B, C, ctx.B_proj_bias, ctx.C_proj_bias,
                                                  batch, L, N)
        delta = delta_r if ctx.checkpoint_lvl == 1 else delta
        u2, delta2, z2, B2, C2, A2, D2, bias2 = BiMambaInnerFn._stack(
            conv1d_out, delta, z, Bv, Cv, A, A_b, D, delta_bias, ctx.is_variable_B, ctx.is_variable_C)
        dout = dout.reshape(batch * L, -1).to(out_proj_weight.dtype)        # ((b l), e)
        dout_y = (out_proj_weight.t() @ dout.t()).view(d_inner, batch, L).transpose(0, 1)
        dconv2, ddelta2, dA2, dB2, dC2, dD2, dbias2, dz2, out_z2 = selective_scan_cuda.bwd(
            u2, delta2, A2, B2, C2, D2, z2, bias2, BiMambaInnerFn._both(dout_y), scan_intermediates, out2, None,
            ctx.delta_softplus, True)
        fold = BiMambaInnerFn._fold
        dxz = torch.empty_like(xz)
        dx, dz = dxz.chunk(2, dim=1)
        dz.copy_(fold(dz2))
        out_z = fold(out_z2)
        dout_proj_weight = dout.t() @ out_z.transpose(1, 2).reshape(batch * L, d_inner)
        dout_proj_bias = dout.sum(0) if not ctx.out_proj_bias_is_None else None
        dA, dA_b = dA2[:d_inner], dA2[d_inner:]
        halves = lambda t: t[:d_inner] + t[d_inner:]
        dx_dbl = torch.empty_like(x_dbl)
        dB_proj_bias = dC_proj_bias = None
        if ctx.is_variable_B:
            dB = fold(dB2).squeeze(1).transpose(1, 2).reshape(batch * L, N)
            dB_proj_bias = dB.sum(0) if ctx.B_proj_bias is not None else None
            dx_dbl[:, R:R + N] = dB
            dB = None
        else:
            dB = halves(dB2)
            dx_dbl[:, R:R + N].zero_()                      # unused columns of the projection: zero gradient
        if ctx.is_variable_C:
            dC = fold(dC2).squeeze(1).transpose(1, 2).reshape(batch * L, N)
            dC_proj_bias = dC.sum(0) if ctx.C_proj_bias is not None else None
            dx_dbl[:, -N:] = dC
            dC = None
        else:
            dC = halves(dC2)
            dx_dbl[:, -N:].zero_()
        ddelta = fold(ddelta2).transpose(0, 1).reshape(d_inner, batch * L)
        ddelta_proj_weight = ddelta @ x_dbl[:, :R]
        dx_dbl[:, :R] = ddelta.t() @ delta_proj_weight
        dconv1d_out = fold(dconv2).transpose(0, 1).reshape(d_inner, batch * L)
        dx_proj_weight = dx_dbl.t() @ conv1d_out.transpose(1, 2).reshape(batch * L, d_inner)
        dconv1d_out = torch.addmm(dconv1d_out, x_proj_weight.t(), dx_dbl.t())
        dconv1d_out = dconv1d_out.view(d_inner, batch, L).transpose(0, 1)
        dx, dconv1d_weight, dconv1d_bias = causal_conv1d_cuda.causal_conv1d_bwd(
            x, conv1d_weight, conv1d_bias, dconv1d_out, dx, True)
        return (dxz, dconv1d_weight.unsqueeze(1), dconv1d_bias if conv1d_bias is not None else None,
                dx_proj_weight, ddelta_proj_weight, dout_proj_weight, dout_proj_bias, dA, dA_b, dB, dC,
                halves(dD2) if D is not None else None, halves(dbias2) if delta_bias is not None else None,
                dB_proj_bias, dC_proj_bias, None, None)


def bimamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias,
                     A, A_b, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None,
                     delta_softplus=True):
    """xz: (batch, 2*d_inner, seqlen) -> (batch, seqlen, d_model); selective_scan_interface.py:616-625."""
    return BiMambaInnerFn.apply(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                                out_proj_bias, A, A_b, B, C, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus)
