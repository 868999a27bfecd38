"""csrc/layernorm.hip (the LayerNorms around the Mamba call, modeling/vivim.py:155-156) against
torch.nn.functional.layer_norm in fp32 -- the operator the reference uses there; through the C ABI (vivim_amd/layernorm.py)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _cm(B, C, L, dtype, dev, gen):
    """(B, L, C) view of (B, C, L) memory, as MambaLayer builds it."""
    return torch.randn(B, C, L, generator=gen).mul_(1.7).add_(0.4).to(dtype).to(dev).transpose(1, 2)


@pytest.mark.parametrize("B,C,L", [(3, 64, 20480), (2, 128, 5120), (2, 320, 1280), (3, 512, 320), (1, 96, 40), (2, 8, 8)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("autocast", [False, True])
@pytest.mark.parametrize("tile", [None, "8", "16", "32"])
def test_layer_norm_channel_major(B, C, L, dtype, autocast, tile, cuda, monkeypatch):
    """Every tile length of the kernels (VIVIM_LN_TT: tokens per wave; None = the automatic choice; a length that does not fit
    the shape -- LDS, or 16 tokens of a 16-bit row of 8-element vectors -- falls back to the next one that does)."""
    from vivim_amd import layernorm as ln
    if tile is None:
        monkeypatch.delenv("VIVIM_LN_TT", raising=False)
    else:
        monkeypatch.setenv("VIVIM_LN_TT", tile)
    gen = torch.Generator().manual_seed(C + L)
    x = _cm(B, C, L, dtype, cuda, gen).requires_grad_(True)
    w = (torch.randn(C, generator=gen) * 0.5 + 1.0).to(cuda).requires_grad_(True)
    b = (torch.randn(C, generator=gen) * 0.3).to(cuda).requires_grad_(True)
    assert ln.supported(x, w)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        y = ln.layer_norm_cm(x, w, b, 1e-5)
        # ATen: f32 under autocast (layer_norm is on its fp32 list), the input's dtype otherwise (it then wants the weights in
        # that dtype too; the fused op keeps them f32 either way)
        want_dtype = F.layer_norm(x.detach(), (C,), w.detach(), b.detach(), 1e-5).dtype if autocast else dtype
    assert y.dtype == want_dtype and y.shape == (B, L, C) and y.is_contiguous()
    g = torch.randn(B, L, C, generator=gen).to(y.dtype).to(cuda)
    y.backward(g)
    # reference: fp32 layer_norm of the same (rounded) inputs, gradient by autograd
    x32 = x.detach().float().requires_grad_(True)
    w32, b32 = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    r = F.layer_norm(x32, (C,), w32, b32, 1e-5)
    r.backward(g.float())
    lo = y.dtype != torch.float32
    assert rel_err(y.float(), r.to(y.dtype).float()) < (4e-3 if lo else 2e-6)
    assert x.grad.shape == x.shape and x.grad.stride() == x.stride()          # dx in x's own (channel-major) layout
    gx_tol = 4e-3 if dtype != torch.float32 else 1e-5
    assert rel_err(x.grad.float(), x32.grad.to(dtype).float()) < gx_tol
    assert rel_err(w.grad, w32.grad) < 1e-4 and rel_err(b.grad, b32.grad) < 1e-4


def test_layer_norm_no_bias_and_unsupported_layouts(cuda):
    from vivim_amd import layernorm as ln
    gen = torch.Generator().manual_seed(2)
    x = _cm(2, 64, 256, torch.float32, cuda, gen)
    w = torch.ones(64, device=cuda)
    y = ln.layer_norm_cm(x, w, None, 1e-6)
    assert rel_err(y, F.layer_norm(x, (64,), w, None, 1e-6)) < 2e-6
    assert not ln.supported(x.contiguous(), w)                                # token-major rows: ATen's kernel is the right one
    assert not ln.supported(_cm(1, 1024, 64, torch.float32, cuda, gen), torch.ones(1024, device=cuda))   # > 512 channels
    assert not ln.supported(_cm(1, 64, 36, torch.bfloat16, cuda, gen), w)     # 36 tokens: not whole 16-byte pieces


def test_mamba_layer_uses_the_fused_norm(cuda, monkeypatch):
    """MambaLayer with and without the fused norm: same output and parameter gradients (the norm's nn.LayerNorm parameters and
    state-dict keys are untouched)."""
    from modeling.vivim import MambaLayer
    from vivim_amd import layernorm as ln
    torch.manual_seed(5)
    layer = MambaLayer(64).to(cuda)
    x = torch.randn(2, 64, 2, 32, 32, device=cuda)          # 4 096 tokens: above layernorm.worthwhile's threshold
    calls = []
    real = ln.layer_norm_cm
    monkeypatch.setattr(ln, "layer_norm_cm", lambda *a: (calls.append(1), real(*a))[1])

    def run():
        for p in layer.parameters():
            p.grad = None
        y = layer(x)
        y.square().mean().backward()
        return y.detach(), {n: p.grad.clone() for n, p in layer.named_parameters()}
    y1, g1 = run()
    assert len(calls) == 2                                   # norm1 and norm2 both took the kernel
    monkeypatch.setenv("VIVIM_NO_FUSED_LAYERNORM", "1")
    y2, g2 = run()
    assert len(calls) == 2
    assert not ln.worthwhile(torch.empty(3, 320, 512, device=cuda)) and ln.worthwhile(torch.empty(3, 5120, 128, device=cuda))
    assert rel_err(y1, y2) < 1e-5
    for n in g1:
        assert rel_err(g1[n], g2[n]) < 2e-4, n
    assert {"norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias"} <= set(g1)
