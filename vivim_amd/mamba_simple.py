"""`Mamba` module with the reference fork's tri-directional "v3" block
(mamba/mamba_ssm/modules/mamba_simple.py:34-264): same constructor arguments, parameter names
(so reference checkpoints load: in_proj, conv1d{,_b,_s}, x_proj{,_b,_s}, dt_proj{,_b,_s},
A{,_b,_s}_log, D{,_b,_s}, out_proj) and forward semantics.

Deliberate differences:
  * `nframes` is honoured per call: the reference hard-codes 5 (mamba_simple.py:54) and its
    chunk/stack re-ordering raises for clips whose token count is not 5 equal chunks; here the
    frame-major -> pixel-major permutation is an exact reshape for any nframes dividing seqlen.
  * `step` / `allocate_inference_cache` (mamba_simple.py:356-413) are the stock single-direction decode step of the
    reference (forward-direction parameters only), here on the single-token HIP kernels; the `inference_params` cache
    plumbing of `forward` (mamba_simple.py:188-200) is not built -- Vivim never passes it.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

import os

from .dirmap import combine_directions, stack_directions
from .causal_conv1d_interface import causal_conv1d_update
from .selective_scan_interface import mamba_inner_fn_no_out_proj, mamba_inner_grouped_fn_no_out_proj
from .selective_state_update import selective_state_update

_DIRECTIONS = ("", "_b", "_s")     # forward in time, backward in time, spatial (pixel-major) order
# what the grouped op takes per direction: (attribute path of the forward direction's parameter, its suffix position)
_FUSED = ("conv1d{}.weight", "conv1d{}.bias", "x_proj{}.weight", "dt_proj{}.weight", "dt_proj{}.bias", "A{}_log", "D{}")


class _FusedViews(torch.autograd.Function):
    """The three directions' parameters of each kind as ONE (3, ...) tensor without a per-step cat / stack: `bufs` are the
    storages the Parameters are views of (Mamba._fuse); the Parameters ride along only to receive the gradients, which go
    back as three views of each incoming one.  One node for all kinds (an `apply` costs ~15 us of host time), no kernel
    on either way."""

    @staticmethod
    def forward(ctx, nb, *args):
        ctx.nb = nb
        return tuple(b.view_as(b) for b in args[:nb])

    @staticmethod
    def backward(ctx, *gs):
        out = [None] * (1 + ctx.nb)
        for g in gs:
            out += [None, None, None] if g is None else [g[0], g[1], g[2]]
        return tuple(out)


class Mamba(nn.Module):
    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False,
                 use_fast_path=True, layer_idx=None, device=None, dtype=None, bimamba_type="none",
                 nframes=5):
        kw = {"device": device, "dtype": dtype}
        super().__init__()
        assert bimamba_type == "v3", "this fork only builds the tri-directional v3 block (mamba_simple.py:125)"
        self.d_model, self.d_state, self.d_conv, self.expand = d_model, d_state, d_conv, expand
        self.d_inner = int(expand * d_model)
        self.dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        self.use_fast_path = use_fast_path
        self.layer_idx = layer_idx
        self.bimamba_type = bimamba_type
        self.nframes = nframes
        self.activation = "silu"

        self.in_proj = nn.Linear(d_model, 2 * self.d_inner, bias=bias, **kw)
        for sfx in _DIRECTIONS:
            self.add_module("conv1d" + sfx, nn.Conv1d(self.d_inner, self.d_inner, kernel_size=d_conv,
                                                      groups=self.d_inner, padding=d_conv - 1,
                                                      bias=conv_bias, **kw))
            self.add_module("x_proj" + sfx, nn.Linear(self.d_inner, self.dt_rank + 2 * d_state, bias=False, **kw))
            dt_proj = nn.Linear(self.dt_rank, self.d_inner, bias=True, **kw)
            self.add_module("dt_proj" + sfx, dt_proj)
            if sfx == "":
                # variance-preserving dt weights and softplus^-1(U_log[dt_min, dt_max]) bias are applied to
                # the forward direction only in the reference (mamba_simple.py:89-108)
                std = self.dt_rank ** -0.5 * dt_scale
                if dt_init == "constant":
                    nn.init.constant_(dt_proj.weight, std)
                elif dt_init == "random":
                    nn.init.uniform_(dt_proj.weight, -std, std)
                else:
                    raise NotImplementedError
                dt = torch.exp(torch.rand(self.d_inner, **kw) * (math.log(dt_max) - math.log(dt_min))
                               + math.log(dt_min)).clamp(min=dt_init_floor)
                with torch.no_grad():
                    dt_proj.bias.copy_(dt + torch.log(-torch.expm1(-dt)))
                dt_proj.bias._no_reinit = True
            # S4D-real A = -(1..N), kept as fp32 log; D = 1 (mamba_simple.py:110-123)
            A_log = torch.log(torch.arange(1, d_state + 1, dtype=torch.float32, device=device)
                              ).repeat(self.d_inner, 1).contiguous()
            a_name = f"A{sfx}_log"
            self.register_parameter(a_name, nn.Parameter(A_log))
            getattr(self, a_name)._no_weight_decay = True
            d_name = "D" + sfx
            self.register_parameter(d_name, nn.Parameter(torch.ones(self.d_inner, device=device)))
            getattr(self, d_name)._no_weight_decay = True
        self.out_proj = nn.Linear(self.d_inner, d_model, bias=bias, **kw)

    def _fuse(self):
        """Make the three directions' parameters of each kind views of one (3, ...) buffer (once; again after a `.to()` /
        `.float()` has replaced the Parameters' storage).  Names, shapes and values of the Parameters do not change, so state
        dicts, optimizers and DDP see what they saw; an optimizer that updates the Parameters in place updates the buffer."""
        def param(path):
            obj = self
            for name in path.split("."):
                obj = getattr(obj, name)
            return obj
        bufs, groups = [], []
        for kind in _FUSED:
            ps = [param(kind.format(sfx)) for sfx in _DIRECTIONS]
            if any(p is None for p in ps):
                bufs.append(None)
                groups.append(None)
                continue
            buf = torch.stack([p.data for p in ps])
            for g, p in enumerate(ps):
                p.data = buf[g]
            bufs.append(buf)
            groups.append(ps)
        self._fused = bufs
        self._fused_live = [b for b in bufs if b is not None]
        self._fused_args = self._fused_live + [p for ps in groups if ps is not None for p in ps]
        # (owner module, attribute, buffer, row) of every fused Parameter: what _fused_params re-checks on each call
        self._fused_where = []
        for kind, buf in zip(_FUSED, bufs):
            if buf is None:
                continue
            for g, sfx in enumerate(_DIRECTIONS):
                *mods, attr = kind.format(sfx).split(".")
                owner = self
                for name in mods:
                    owner = getattr(owner, name)
                # (the row's address, dtype and device are kept as plain values: a `buf[g]` view per check costs microseconds)
                self._fused_where.append((owner._parameters, attr, buf[g].data_ptr(), buf.dtype, buf.device))

    def _fused_params(self):
        """-> [conv_w (3, D, 1, W), conv_b (3, D) | None, x_proj_w (3, R + 2N, D), dt_proj_w (3, D, R), dt_bias (3, D),
        A_log (3, D, N), D (3, D)] with autograd edges to the per-direction Parameters.
        EVERY fused Parameter is looked up again and compared with its row of the buffer (21 registry reads and data_ptr()
        calls per forward, about 20 us): a Parameter that was re-assigned, or whose .data got new storage (`m.A_b_log.data = t`, a partial
        load_state_dict(assign=True), a `.to()`), makes the module fuse again instead of training on a buffer that
        state_dict() no longer reports."""
        where = getattr(self, "_fused_where", None)
        ok = where is not None
        if ok:
            args = self._fused_args
            k = len(self._fused_live)
            for i, (params, attr, ptr, dtype, device) in enumerate(where):
                prm = params.get(attr)                   # the module's own registry: what getattr(owner, attr) resolves to
                if prm is not args[k + i] or prm.data_ptr() != ptr or prm.dtype != dtype or prm.device != device:
                    ok = False
                    break
        if not ok:
            self._fuse()
        bufs = self._fused
        views = iter(_FusedViews.apply(len(self._fused_live), *self._fused_args))
        return [None if b is None else next(views) for b in bufs]

    def _inner(self, xz, sfx):
        conv, x_proj, dt_proj = (getattr(self, n + sfx) for n in ("conv1d", "x_proj", "dt_proj"))
        A = -torch.exp(getattr(self, f"A{sfx}_log").float())
        return mamba_inner_fn_no_out_proj(
            xz, conv.weight, conv.bias, x_proj.weight, dt_proj.weight, A, None, None,
            getattr(self, "D" + sfx).float(), delta_bias=dt_proj.bias.float(), delta_softplus=True)

    def forward(self, hidden_states, inference_params=None, nframes=None):
        """hidden_states: (B, L, d_model) with L = nframes * H * W in frame-major order -> same shape."""
        if inference_params is not None:
            raise NotImplementedError("decode-time inference cache is outside Vivim's path")
        batch, seqlen, _ = hidden_states.shape
        nf = self.nframes if nframes is None else nframes
        if seqlen % nf != 0:
            raise ValueError(f"seqlen {seqlen} is not a multiple of nframes {nf}")
        # in_proj and the BLD -> (B, 2*d_inner, L) transpose in one matmul; xz strides are (L, B*L, 1)
        xz = (self.in_proj.weight @ hidden_states.reshape(batch * seqlen, -1).t()
              ).view(2 * self.d_inner, batch, seqlen).transpose(0, 1)
        if self.in_proj.bias is not None:
            xz = xz + self.in_proj.bias.to(xz.dtype)[:, None]
        hw = seqlen // nf
        # the direction-map kernels move whole 16-byte vectors: token counts that are not a multiple of 8 (never the
        # case for Vivim's H*W = 64 * k) take the reference's three-call composition with torch flips / permutes
        if os.environ.get("VIVIM_SEPARATE_DIRECTIONS", "0") == "1" or seqlen % 8 != 0:
            return self._forward_separate(xz, batch, seqlen, nf, hw)
        # The three directions side by side on the channel axis -> ONE conv1d launch and ONE scan launch
        # (n_groups = 3) per block instead of three of each (selective_scan_interface.MambaInnerGroupedFnNoOutProj).
        D = self.d_inner
        xz3 = stack_directions(xz, nf)                    # (B, 2, 3, D, L): identity / flip / frame interleave, one read
        # the three directions' parameters as (3, ...) views of persistent fused storage: no per-step cat / stack
        conv_w, conv_b, x_proj_w, dt_proj_w, dt_bias, A_log, Dp = self._fused_params()
        A = -torch.exp(A_log.float()).view(3 * D, self.d_state)
        o3 = mamba_inner_grouped_fn_no_out_proj(
            xz3, conv_w.view(3 * D, self.d_conv), conv_b.view(3 * D) if conv_b is not None else None, x_proj_w, dt_proj_w,
            A, Dp.float().view(3 * D), dt_bias.float().view(3 * D), True).view(batch, 3, D, seqlen)
        y = combine_directions(o3, nf).transpose(1, 2)    # (out + out_b.flip + out_s^-1) / 3, one write
        return F.linear(y, self.out_proj.weight, self.out_proj.bias)

    def _forward_separate(self, xz, batch, seqlen, nf, hw):
        """The reference's call pattern: three `mamba_inner_fn_no_out_proj` calls (mamba_simple.py:220-262)."""
        out = self._inner(xz, "")
        out_b = self._inner(xz.flip([-1]), "_b").flip([-1])
        xz_s = xz.reshape(batch, 2 * self.d_inner, nf, hw).transpose(2, 3).reshape(batch, 2 * self.d_inner, seqlen)
        out_s = self._inner(xz_s, "_s")
        out_s = out_s.reshape(batch, self.d_inner, hw, nf).transpose(2, 3).reshape(batch, self.d_inner, seqlen)
        y = (out + out_b + out_s).transpose(1, 2) / 3
        return F.linear(y, self.out_proj.weight, self.out_proj.bias)

    def step(self, hidden_states, conv_state, ssm_state):
        """One token through the forward-direction recurrence (mamba_simple.py:356-399): hidden_states (B, 1, d_model),
        conv_state (B, d_inner, d_conv) and ssm_state (B, d_inner, d_state) advanced in place
        -> (out (B, 1, d_model), conv_state, ssm_state)."""
        assert hidden_states.shape[1] == 1, "Only support decoding with 1 token at a time for now"
        xz = self.in_proj(hidden_states.squeeze(1))                       # (B, 2D)
        x, z = xz.chunk(2, dim=-1)
        x = causal_conv1d_update(x.contiguous(), conv_state, self.conv1d.weight.squeeze(1), self.conv1d.bias, self.activation)
        x_db = self.x_proj(x)                                             # (B, dt_rank + 2 * d_state)
        dt, B, C = torch.split(x_db, [self.dt_rank, self.d_state, self.d_state], dim=-1)
        dt = F.linear(dt, self.dt_proj.weight)                            # the bias goes in with the softplus below
        A = -torch.exp(self.A_log.float())
        y = selective_state_update(ssm_state, x, dt, A, B, C, self.D, z=z, dt_bias=self.dt_proj.bias, dt_softplus=True)
        return self.out_proj(y).unsqueeze(1), conv_state, ssm_state

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        """Zero conv / ssm states for `step` (mamba_simple.py:401-413)."""
        device = self.out_proj.weight.device
        conv_state = torch.zeros(batch_size, self.d_inner, self.d_conv, device=device,
                                 dtype=self.conv1d.weight.dtype if dtype is None else dtype)
        ssm_state = torch.zeros(batch_size, self.d_inner, self.d_state, device=device,
                                dtype=self.dt_proj.weight.dtype if dtype is None else dtype)
        return conv_state, ssm_state
