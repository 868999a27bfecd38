"""Vivim model modules with the reference's signatures (modeling/vivim.py:28-348): LayerNorm, DWConv, Mlp,
MambaLayer, mamba_block, Vivim.  Only the Temporal Mamba Block inside MambaLayer runs our kernels; the
SegFormer encoder stages, the depthwise Conv3d MLP and the decode head stay on stock PyTorch-ROCm.

Differences from the reference, all additive:
  * MambaLayer passes the clip length it sees (x.shape[2]) to Mamba, so clip_length != 5 works
    (the reference leaves Mamba.nframes at its hard-coded 5, vivim.py:116-123 / mamba_simple.py:54);
    d_state / d_conv / expand can be set through `mamba_block(..., mamba_kwargs=...)`.
  * `Vivim(..., backbone=None)`: the reference always downloads
    "nvidia/segformer-b3-finetuned-ade-512-512" (vivim.py:264).  With `backbone=None` we do the same;
    offline, pass `backbone=segformer_b3_random()` (same architecture from a local SegformerConfig,
    random init) -- what bench.py and the tests do.
  * works with both SegFormer module layouts of `transformers` (4.x `encoder.patch_embeddings/block/
    layer_norm` + `decode_head.linear_c`, and 5.x `stages[i].{patch_embeddings,blocks,layer_norm}` +
    `decode_head.linear_projections`).
  * the Mlp's depthwise Conv3d runs on the token-major HIP kernel (csrc/dwconv.hip) when the
    tensor qualifies; `fast_backbone_dwconv=True` swaps the SegFormer blocks' DropPath for the one-kernel form and routes the SegFormer Mix-FFN 3x3 depthwise convs
    through the same kernel (same parameters, same math, state-dict keys unchanged).
  * timm is not required: DropPath / trunc_normal_ are the torch equivalents.
"""
import math

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dwconv as _dw
from . import layernorm as _ln
from .mamba_simple import Mamba


class DropPath(nn.Module):
    """Stochastic depth per sample (timm.models.layers.DropPath semantics)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        # timm: random_tensor.bernoulli_(keep).div_(keep); x * random_tensor -- the scaling stays on the (B, 1, ...) mask,
        # the activation is touched by ONE elementwise kernel forward and one backward
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep).div_(keep)
        return x * mask


def _init_weights(m):
    # modeling/vivim.py:83-96 / 132-145
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=0.02)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)
    elif isinstance(m, nn.Conv2d):
        fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
        m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
        if m.bias is not None:
            m.bias.data.zero_()


class LayerNorm(nn.Module):
    """channels_last (default) or channels_first LayerNorm (modeling/vivim.py:28-54)."""

    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_last"):
        super().__init__()
        if data_format not in ("channels_last", "channels_first"):
            raise NotImplementedError
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps
        self.data_format = data_format
        self.normalized_shape = (normalized_shape,)

    def forward(self, x):
        if self.data_format == "channels_last":
            return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
        mean = x.mean(1, keepdim=True)
        var = (x - mean).pow(2).mean(1, keepdim=True)
        x = (x - mean) / torch.sqrt(var + self.eps)
        return self.weight[:, None, None] * x + self.bias[:, None, None]


class DWConv(nn.Module):
    """3x3x3 depthwise Conv3d over (frames, H, W) on a token sequence (modeling/vivim.py:57-68)."""

    def __init__(self, dim=768):
        super().__init__()
        self.dwconv = nn.Conv3d(dim, dim, 3, 1, 1, bias=True, groups=dim)

    def forward(self, x, nf, H, W):
        B, _, C = x.shape
        if _dw.supported(x, self.dwconv.weight):
            # token-major HIP kernels (csrc/dwconv.hip): same math, no transposes, no MIOpen naive conv
            return _dw.depthwise_conv_tokens(x, self.dwconv.weight, self.dwconv.bias, nf, H, W)
        x = self.dwconv(x.transpose(1, 2).view(B, C, nf, H, W))
        return x.flatten(2).transpose(1, 2)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.dwconv = DWConv(hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)
        self.apply(_init_weights)

    def forward(self, x, nf, H, W):
        x = self.fc1(x)
        conv = self.dwconv.dwconv
        if isinstance(self.act, nn.GELU) and self.act.approximate == "none" and _dw.supported(x, conv.weight) \
                and not os.environ.get("VIVIM_NO_DWCONV_GELU"):
            # act(dwconv(.)) as one kernel: GELU in the convolution's epilogue (csrc/dwconv.hip, act = 1)
            x = self.drop(_dw.depthwise_conv_gelu_tokens(x, conv.weight, conv.bias, nf, H, W))
        else:
            x = self.drop(self.act(self.dwconv(x, nf, H, W)))
        return self.drop(self.fc2(x))


class MambaLayer(nn.Module):
    """LN -> tri-directional Mamba -> DropPath residual; LN -> Mlp(Conv3d) -> DropPath residual
    (modeling/vivim.py:111-159)."""

    def __init__(self, dim, d_state=16, d_conv=4, expand=2, mlp_ratio=4, drop=0.0, drop_path=0.0,
                 act_layer=nn.GELU):
        super().__init__()
        self.dim = dim
        self.norm1 = nn.LayerNorm(dim)
        self.mamba = Mamba(d_model=dim, d_state=d_state, d_conv=d_conv, expand=expand, bimamba_type="v3")
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.apply(_init_weights)

    @staticmethod
    def _norm(norm, x_flat):
        """nn.LayerNorm's parameters, evaluated by csrc/layernorm.hip when x_flat is the channel-major view built below (the
        transpose is then the kernel's read pattern instead of a copy in front of ATen's row kernel) and has enough tokens for
        that to pay (layernorm.worthwhile); ATen otherwise."""
        if (_ln.supported(x_flat, norm.weight) and _ln.worthwhile(x_flat)
                and not os.environ.get("VIVIM_NO_FUSED_LAYERNORM")):
            return _ln.layer_norm_cm(x_flat, norm.weight, norm.bias, norm.eps)
        return norm(x_flat)

    def forward(self, x):
        B, C, nf, H, W = x.shape
        assert C == self.dim
        x_flat = x.reshape(B, C, nf * H * W).transpose(-1, -2)          # frame-major tokens (vivim.py:151-153)
        x_flat = x_flat + self.drop_path(self.mamba(self._norm(self.norm1, x_flat), nframes=nf))
        x_flat = x_flat + self.drop_path(self.mlp(self._norm(self.norm2, x_flat), nf, H, W))
        return x_flat.transpose(-1, -2).reshape(B, C, nf, H, W)


def _encoder_parts(backbone):
    """-> (patch_embeddings, blocks, layer_norms) per stage for either transformers layout."""
    seg = backbone.segformer
    if hasattr(seg, "encoder"):                                      # transformers 4.x
        enc = seg.encoder
        return list(enc.patch_embeddings), list(enc.block), list(enc.layer_norm)
    return ([s.patch_embeddings for s in seg.stages], [s.blocks for s in seg.stages],
            [s.layer_norm for s in seg.stages])


class _TokenDWConv2d(nn.Module):
    """Drop-in for transformers' SegformerDepthWiseConv (forward(hidden_states, height, width) on (B, H*W, C)
    tokens): same `dwconv` nn.Conv2d parameters, but evaluated by the token-major HIP kernel (csrc/dwconv.hip)
    instead of transpose -> MIOpen depthwise conv -> transpose.  Opt-in (mamba_block(fast_backbone_dwconv=True))."""

    def __init__(self, hf_module):
        super().__init__()
        self.dwconv = hf_module.dwconv

    def forward(self, hidden_states, height, width):
        if _dw.supported(hidden_states, self.dwconv.weight):
            return _dw.depthwise_conv_tokens(hidden_states, self.dwconv.weight, self.dwconv.bias, 1, height, width)
        b, _, c = hidden_states.shape
        y = self.dwconv(hidden_states.transpose(1, 2).view(b, c, height, width))
        return y.flatten(2).transpose(1, 2)


def _swap_backbone_dwconv(module):
    """Replace every 3x3 depthwise conv wrapper inside the SegFormer Mix-FFNs (attribute `dwconv` holding a
    module that itself owns a depthwise nn.Conv2d called `dwconv`)."""
    n = 0
    for m in module.modules():
        inner = getattr(m, "dwconv", None)
        conv = getattr(inner, "dwconv", None)
        if isinstance(conv, nn.Conv2d) and conv.groups == conv.in_channels and conv.kernel_size == (3, 3) \
                and conv.stride == (1, 1) and conv.padding == (1, 1) and not isinstance(inner, _TokenDWConv2d):
            m.dwconv = _TokenDWConv2d(inner)
            n += 1
    return n


def _swap_backbone_droppath(module):
    """Replace the SegFormer blocks' stochastic-depth modules (transformers' SegformerDropPath: rand, add, floor on the
    mask, then a full-size div and a full-size mul) by `DropPath` above: the same keep/drop distribution and scaling with
    one full-size kernel instead of two and three launches instead of five (58 of them per step forward, and again
    backward, in a step whose critical path is the host)."""
    n = 0
    for m in module.modules():
        dp = getattr(m, "drop_path", None)
        if isinstance(dp, nn.Module) and type(dp).__name__ == "SegformerDropPath":
            m.drop_path = DropPath(float(dp.drop_prob))
            n += 1
    return n


class _Encoder(nn.Module):
    """Holds the SegFormer encoder pieces under the reference's attribute names."""

    def __init__(self, backbone):
        super().__init__()
        pe, blocks, norms = _encoder_parts(backbone)
        self.patch_embeddings = nn.ModuleList(pe)
        self.block = nn.ModuleList(blocks)
        self.layer_norm = nn.ModuleList(norms)


def _run_block(blk, hs, height, width):
    out = blk(hs, height, width)
    return out[0] if isinstance(out, (tuple, list)) else out


class mamba_block(nn.Module):
    """SegFormer encoder stages interleaved with MambaLayer stages (modeling/vivim.py:163-231)."""

    def __init__(self, backbone, in_chans=1, depths=[2, 2, 2, 2], dims=[64, 128, 320, 512],
                 drop_path_rate=0.0, layer_scale_init_value=1e-6, out_indices=[0, 1, 2, 3], mamba_kwargs=None,
                 fast_backbone_dwconv=False):
        super().__init__()
        self.downsample_layers = _Encoder(backbone)
        if fast_backbone_dwconv:
            _swap_backbone_dwconv(self.downsample_layers)
            _swap_backbone_droppath(self.downsample_layers)
        dp_rates = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        mk = mamba_kwargs or {}
        self.stages = nn.ModuleList()
        for i in range(len(dims)):
            # NB the reference indexes dp_rates by STAGE, not by layer (vivim.py:186); kept.
            self.stages.append(nn.Sequential(*[nn.Sequential(MambaLayer(dim=dims[i], drop_path=dp_rates[i], **mk))
                                               for _ in range(depths[i])]))
        self.out_indices = out_indices

    def forward_features(self, x):
        bz, nf = x.shape[:2]
        hs = x.reshape(bz * nf, *x.shape[-3:])
        outs = []
        enc = self.downsample_layers
        for embed, blocks, _norm, mam_stage in zip(enc.patch_embeddings, enc.block, enc.layer_norm, self.stages):
            hs, height, width = embed(hs)
            for blk in blocks:
                hs = _run_block(blk, hs, height, width)
            # (the stage layer norm is skipped on purpose, vivim.py:211-212)
            hs = hs.reshape(bz * nf, height, width, -1).permute(0, 3, 1, 2).contiguous()
            hs = hs.reshape(bz, nf, *hs.shape[-3:]).transpose(1, 2)             # (B, C, nf, H, W)
            hs = mam_stage(hs)                                                   # (B, C, nf, H, W)
            hs = hs.transpose(1, 2).reshape(bz * nf, hs.shape[1], height, width)
            outs.append(hs)
        return tuple(outs)

    def forward(self, x):
        return self.forward_features(x)


def segformer_b3_config(num_labels=150):
    from transformers import SegformerConfig
    return SegformerConfig(num_channels=3, num_encoder_blocks=4, depths=[3, 4, 18, 3], sr_ratios=[8, 4, 2, 1],
                           hidden_sizes=[64, 128, 320, 512], patch_sizes=[7, 3, 3, 3], strides=[4, 2, 2, 2],
                           num_attention_heads=[1, 2, 5, 8], mlp_ratios=[4, 4, 4, 4], decoder_hidden_size=768,
                           num_labels=num_labels)


def segformer_b3_random(num_labels=150):
    """SegFormer-b3 (the architecture of nvidia/segformer-b3-finetuned-ade-512-512) with random weights,
    built from a local config -- no hub access."""
    from transformers import SegformerForSemanticSegmentation
    return SegformerForSemanticSegmentation(segformer_b3_config(num_labels))


class Vivim(nn.Module):
    """modeling/vivim.py:234-348.  forward(x_in[B, nf, 3, H, W]) -> logits[B*nf, out_chans, H, W]
    (+ edge map when with_edge)."""

    def __init__(self, in_chans=3, out_chans=3, depths=[2, 2, 2, 2], feat_size=[64, 128, 320, 512],
                 drop_path_rate=0.2, layer_scale_init_value=1e-6, hidden_size: int = 768, norm_name="instance",
                 conv_block: bool = True, res_block: bool = True, spatial_dims=2, with_edge=False,
                 dropout_rate=0.3, backbone=None, mamba_kwargs=None, fast_backbone_dwconv=False) -> None:
        super().__init__()
        self.hidden_size = hidden_size
        self.in_chans, self.out_chans = in_chans, out_chans
        self.depths, self.feat_size = depths, feat_size
        self.drop_path_rate = drop_path_rate
        self.layer_scale_init_value = layer_scale_init_value
        self.dropout_rate = dropout_rate
        self.spatial_dims = spatial_dims
        if backbone is None:
            from transformers import SegformerForSemanticSegmentation
            backbone = SegformerForSemanticSegmentation.from_pretrained("nvidia/segformer-b3-finetuned-ade-512-512")
        self.encoder = mamba_block(backbone, in_chans, depths=depths, dims=feat_size,
                                   drop_path_rate=drop_path_rate, mamba_kwargs=mamba_kwargs,
                                   fast_backbone_dwconv=fast_backbone_dwconv)
        self.decoder = backbone.decode_head
        self.feature_dropout = nn.Dropout2d(dropout_rate)
        self.out = nn.Conv2d(768, out_chans, kernel_size=1)
        self.with_edge = with_edge
        if with_edge:
            self.edgeocr_cls_head = nn.Conv2d(64, 1, kernel_size=1, stride=1, padding=0, bias=True)

    def _decoder_projections(self):
        dec = self.decoder
        return dec.linear_c if hasattr(dec, "linear_c") else dec.linear_projections

    def decode(self, encoder_hidden_states, bz, nf):
        batch_size = encoder_hidden_states[-1].shape[0]
        size0 = encoder_hidden_states[0].shape[2:]
        feats = ()
        for state, mlp in zip(encoder_hidden_states, self._decoder_projections()):
            height, width = state.shape[2], state.shape[3]
            state = mlp(state).permute(0, 2, 1).reshape(batch_size, -1, height, width)
            state = F.interpolate(state, size=size0, mode="bilinear", align_corners=False)
            if torch.rand(1).item() > 0.5:          # per-map dropout coin flip on the CPU RNG (vivim.py:310-312)
                state = F.dropout(state, p=self.dropout_rate / 2, training=self.training)
            feats += (state,)
        # linear_fuse / out stay nn.Conv2d on MIOpen, as in the reference.  They were batched GEMMs for a while (+18%
        # frames/s): the library GEMM for 768 x 3072 x 61440 bf16 on the channels-last concat (hipBLASLt picks
        # Custom_Cijk_Alik_Bljk_BBS_BH_Bias_HA_S_SAV_NTD_SK3_UserArgs_MT256x256x64) dies with a GPU memory access fault
        # in no_grad forwards and, with other allocation sizes, in training; no BLAS backend / workspace setting
        # avoided it.
        hidden = self.decoder.linear_fuse(torch.cat(feats[::-1], dim=1))
        hidden = self.decoder.activation(self.decoder.batch_norm(hidden))
        hidden = self.decoder.dropout(self.decoder.dropout(hidden))   # applied twice (vivim.py:319-322)
        hidden = self.feature_dropout(hidden)
        return self.out(hidden)

    def forward(self, x_in):
        bz, nf, nc, h, w = x_in.shape
        outs = self.encoder(x_in)
        logits = F.interpolate(self.decode(outs, bz, nf), size=(h, w), mode="bilinear", align_corners=False)
        if self.with_edge:
            edge = F.interpolate(self.edgeocr_cls_head(outs[0]), size=(h, w), mode="bilinear", align_corners=False)
            return logits, edge
        return logits
