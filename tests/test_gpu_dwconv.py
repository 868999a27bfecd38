"""GPU: token-major depthwise conv kernels (csrc/dwconv.hip) against the op they replace --
the reference's nn.Conv3d(groups=C) on the transposed view (modeling/vivim.py:57-68) -- evaluated by PyTorch in
fp32 (floating-point kernel: a torch fp32 reference is the oracle here).  Tolerance: rel-err <= 1e-3 for 16-bit
I/O against the rounded fp32 result, <= 2e-5 for fp32 I/O."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _ref(x, weight, bias, D, H, W):
    """fp32 torch reference on the (B, C, D, H, W) view, exactly DWConv.forward's math."""
    B, L, C = x.shape
    xf = x.float().transpose(1, 2).reshape(B, C, D, H, W)
    if weight.dim() == 4:
        y = F.conv2d(xf.reshape(B, C, H, W), weight.float(), None if bias is None else bias.float(), padding=1, groups=C)
    else:
        y = F.conv3d(xf, weight.float(), None if bias is None else bias.float(), padding=1, groups=C)
    return y.reshape(B, C, L).transpose(1, 2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,D,H,W,C,k3", [(2, 5, 8, 8, 64, True), (1, 3, 7, 10, 24, True), (3, 1, 16, 16, 256, False),
                                          (2, 1, 5, 3, 8, False), (1, 8, 4, 4, 2048, True), (2, 5, 1, 1, 16, True)])
def test_dwconv_matches_torch(dtype, B, D, H, W, C, k3, cuda):
    from vivim_amd.dwconv import depthwise_conv_tokens, supported
    g = torch.Generator().manual_seed(B * 100 + C)
    x = torch.randn(B, D * H * W, C, generator=g).to(dtype).to(cuda).requires_grad_(True)
    wshape = (C, 1, 3, 3, 3) if k3 else (C, 1, 3, 3)
    w = (torch.randn(*wshape, generator=g) * 0.3).to(cuda).requires_grad_(True)
    b = torch.randn(C, generator=g).to(cuda).requires_grad_(True)
    dy = torch.randn(B, D * H * W, C, generator=g).to(dtype).to(cuda)
    if not supported(x, w):
        pytest.skip("shape outside the kernel's alignment envelope (torch path is used)")
    y = depthwise_conv_tokens(x, w, b, D, H, W)
    y.backward(dy)
    got = (y.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone())
    x.grad = w.grad = b.grad = None
    yr = _ref(x, w, b, D, H, W)
    yr.backward(dy.float())
    tol = 2e-5 if dtype == torch.float32 else 1e-3
    assert y.dtype == dtype and y.shape == x.shape
    assert rel_err(got[0].float(), yr.detach().to(dtype).float()) < tol
    assert rel_err(got[1].float(), x.grad.to(dtype).float()) < tol
    assert rel_err(got[2], w.grad) < (1e-4 if dtype == torch.float32 else 2e-3)
    assert rel_err(got[3], b.grad) < (1e-4 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,D,H,W,C,k3", [(2, 5, 8, 8, 64, True), (1, 3, 7, 10, 24, True), (3, 1, 16, 16, 256, False),
                                          (1, 8, 4, 4, 2048, True)])
def test_dwconv_gelu_epilogue_matches_torch(dtype, B, D, H, W, C, k3, cuda):
    """gelu(dwconv(x)) with the activation in the convolution's epilogue and the convolution recomputed in the backward
    (vivim_dwconv_params.act = 1 / 2) against F.gelu(conv3d(x)) in fp32 (the Mlp's act(dwconv(fc1 x)), vivim.py:99-106)."""
    from vivim_amd.dwconv import depthwise_conv_gelu_tokens, supported
    g = torch.Generator().manual_seed(B * 100 + C + 1)
    x = torch.randn(B, D * H * W, C, generator=g).to(dtype).to(cuda).requires_grad_(True)
    wshape = (C, 1, 3, 3, 3) if k3 else (C, 1, 3, 3)
    w = (torch.randn(*wshape, generator=g) * 0.3).to(cuda).requires_grad_(True)
    b = torch.randn(C, generator=g).to(cuda).requires_grad_(True)
    dy = torch.randn(B, D * H * W, C, generator=g).to(dtype).to(cuda)
    if not supported(x, w):
        pytest.skip("shape outside the kernel's alignment envelope (torch path is used)")
    y = depthwise_conv_gelu_tokens(x, w, b, D, H, W)
    y.backward(dy)
    got = (y.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone())
    x.grad = w.grad = b.grad = None
    yr = F.gelu(_ref(x, w, b, D, H, W))
    yr.backward(dy.float())
    tol = 2e-5 if dtype == torch.float32 else 1e-3
    assert y.dtype == dtype and y.shape == x.shape
    assert rel_err(got[0].float(), yr.detach().to(dtype).float()) < tol
    # the gradient passes through a value the kernel rounds to the I/O dtype (dy * gelu'): 16-bit bound as for the plain op
    assert rel_err(got[1].float(), x.grad.to(dtype).float()) < (tol if dtype == torch.float32 else 3e-3)
    assert rel_err(got[2], w.grad) < (1e-4 if dtype == torch.float32 else 3e-3)
    assert rel_err(got[3], b.grad) < (1e-4 if dtype == torch.float32 else 3e-3)


def test_mlp_uses_the_fused_activation(cuda, monkeypatch):
    """Mlp (vivim.py:86-108) with nn.GELU: fused path == dwconv followed by nn.GELU, forward and gradients."""
    from modeling.vivim import Mlp
    torch.manual_seed(0)
    m = Mlp(64, 256).to(cuda)
    x = torch.randn(2, 3 * 8 * 8, 64, device=cuda, requires_grad=True)

    def run():
        for p in m.parameters():
            p.grad = None
        x.grad = None
        y = m(x, 3, 8, 8)
        y.square().mean().backward()
        return y.detach(), x.grad.clone(), [p.grad.clone() for p in m.parameters()]
    y1, dx1, g1 = run()
    monkeypatch.setenv("VIVIM_NO_DWCONV_GELU", "1")
    y2, dx2, g2 = run()
    assert rel_err(y1, y2) < 2e-5 and rel_err(dx1, dx2) < 1e-4
    for a, c in zip(g1, g2):
        assert rel_err(a, c) < 1e-4


def test_dwconv_module_path_matches_conv3d_module(cuda):
    """DWConv (vivim.py:57-68) gives the same result through the HIP kernels as through nn.Conv3d."""
    from modeling.vivim import DWConv
    torch.manual_seed(0)
    m = DWConv(128).to(cuda)
    x = torch.randn(2, 5 * 8 * 8, 128, device=cuda)
    y = m(x, 5, 8, 8)
    yr = m.dwconv(x.transpose(1, 2).reshape(2, 128, 5, 8, 8)).flatten(2).transpose(1, 2)
    assert rel_err(y, yr) < 2e-5


def test_dwconv_full_size(cuda):
    """Stage-0 Mlp shape of the benchmarked config (B 3, nf 5, 64x64, C 256, bf16): adjointness
    <conv(x), g> == <x, conv_T(g)> and the slice-vs-torch check."""
    from vivim_amd.dwconv import depthwise_conv_tokens
    g = torch.Generator().manual_seed(1)
    B, D, H, W, C = 3, 5, 64, 64, 256
    x = torch.randn(B, D * H * W, C, generator=g).to(torch.bfloat16).to(cuda).requires_grad_(True)
    w = (torch.randn(C, 1, 3, 3, 3, generator=g) * 0.2).to(cuda).requires_grad_(True)
    dy = torch.randn(B, D * H * W, C, generator=g).to(torch.bfloat16).to(cuda)
    y = depthwise_conv_tokens(x, w, None, D, H, W)
    y.backward(dy)
    sl = slice(0, 16)
    yr = _ref(x[:1, :, sl].detach(), w[sl].detach(), None, D, H, W)
    assert rel_err(y[:1, :, sl].float(), yr.to(torch.bfloat16).float()) < 1e-3
    lhs = (y.detach().double() * dy.double()).sum()
    rhs = (x.detach().double() * x.grad.double()).sum()
    assert abs(float(lhs - rhs)) / abs(float(lhs)) < 2e-2      # both sides carry bf16 rounding of y and dx


def test_dwconv_random_cases(cuda):
    """Hypothesis-drawn geometries (depth 1..8, odd heights / widths, channel counts from 2 to 512, 2-D and 3-D taps,
    with and without bias, all dtypes) against the fp32 torch convolution, forward and all three gradients."""
    hyp = pytest.importorskip("hypothesis")
    st = hyp.strategies
    from vivim_amd.dwconv import depthwise_conv_tokens, supported

    @hyp.settings(max_examples=60, deadline=None, derandomize=True, suppress_health_check=list(hyp.HealthCheck))
    @hyp.given(B=st.integers(1, 3), D=st.integers(1, 8), H=st.integers(1, 12), W=st.integers(1, 12),
               C=st.sampled_from([2, 8, 24, 64, 130, 512]), k3=st.booleans(), has_bias=st.booleans(),
               dtype=st.sampled_from([torch.float32, torch.bfloat16, torch.float16]), seed=st.integers(0, 999))
    def run(B, D, H, W, C, k3, has_bias, dtype, seed):
        if not k3:
            D = 1
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(B, D * H * W, C, generator=g).to(dtype).to(cuda).requires_grad_(True)
        w = (torch.randn(*((C, 1, 3, 3, 3) if k3 else (C, 1, 3, 3)), generator=g) * 0.3).to(cuda).requires_grad_(True)
        b = torch.randn(C, generator=g).to(cuda).requires_grad_(True) if has_bias else None
        if not supported(x, w):
            return
        dy = torch.randn(B, D * H * W, C, generator=g).to(dtype).to(cuda)
        y = depthwise_conv_tokens(x, w, b, D, H, W)
        y.backward(dy)
        got = (y.detach(), x.grad.clone(), w.grad.clone(), None if b is None else b.grad.clone())
        x.grad = w.grad = None
        if b is not None:
            b.grad = None
        yr = _ref(x, w, b, D, H, W)
        yr.backward(dy.float())
        tol = 2e-5 if dtype == torch.float32 else 1e-3
        assert rel_err(got[0].float(), yr.detach().to(dtype).float()) < tol
        assert rel_err(got[1].float(), x.grad.to(dtype).float()) < tol
        assert rel_err(got[2], w.grad) < (1e-4 if dtype == torch.float32 else 3e-3)
        if b is not None:
            assert rel_err(got[3], b.grad) < (1e-4 if dtype == torch.float32 else 3e-3)

    run()
