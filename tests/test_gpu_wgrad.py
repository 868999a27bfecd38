"""csrc/wgrad.hip (the weight-gradient products of the fused inner op's backward, selective_scan_interface.py:273, 276) against
torch's fp32 bmm of the same (rounded) operands; through the C ABI (vivim_amd/wgrad.py)."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("G,M,N,K", [(3, 128, 4, 61440), (3, 36, 128, 61440), (3, 256, 8, 15360), (3, 40, 256, 15360),
                                     (1, 4, 4, 8), (2, 130, 17, 4104), (1, 64, 64, 256), (3, 1024, 32, 960), (2, 16, 300, 2056)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_wgrad_nt_matches_fp32_bmm(G, M, N, K, dtype, cuda, monkeypatch):
    from vivim_amd import wgrad
    monkeypatch.setattr(wgrad, "MIN_TOKENS", 0)
    gen = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randn(G, M, K, generator=gen).to(dtype).to(cuda)
    b = torch.randn(G, N, K, generator=gen).to(dtype).to(cuda)
    assert wgrad.supported(a, b)
    out = wgrad.wgrad_nt(a, b)
    ref = torch.bmm(a.float(), b.float().transpose(1, 2))
    assert out.dtype == torch.float32 and out.shape == (G, M, N)
    assert rel_err(out, ref) < 2e-6                       # exact products, f32 sums in another order


def test_wgrad_nt_row_slices_and_refusals(cuda):
    """The operands the grouped op passes: x_dbl[:, :R] is a row slice of (G, R + 2N, K); f32 and unaligned operands are the
    caller's (torch.bmm) business."""
    from vivim_amd import _lib, wgrad
    gen = torch.Generator().manual_seed(3)
    G, D, R, N, K = 3, 128, 4, 16, 10240
    x_dbl = torch.randn(G, R + 2 * N, K, generator=gen).to(torch.bfloat16).to(cuda)
    ddelta = torch.randn(G, D, K, generator=gen).to(torch.bfloat16).to(cuda)
    x_r = x_dbl[:, :R]
    assert wgrad.supported(ddelta, x_r)
    out = wgrad.wgrad_nt(ddelta, x_r)
    assert rel_err(out, torch.bmm(ddelta.float(), x_r.float().transpose(1, 2))) < 2e-6
    assert not wgrad.supported(ddelta.float(), x_r.float())
    assert not wgrad.supported(ddelta[:, :, 4:], x_r[:, :, 4:])          # rows no longer 16-byte aligned
    assert not wgrad.supported(ddelta[:, :, :9004], x_r[:, :, :9004])    # K not a multiple of 8
    assert not wgrad.supported(ddelta[:, :, :4096], x_r[:, :, :4096])    # few tokens: the library GEMM is as fast
    P = _lib.WgradNtParams()
    P.groups, P.m, P.n, P.k, P.itype = 1, 4, 4, 8, _lib.F32
    a = torch.zeros(1, 4, 8, device=cuda)
    P.a = P.b = P.out = a.data_ptr()
    P.a_row_stride = P.b_row_stride = 8
    P.out_row_stride = 4
    with pytest.raises(RuntimeError, match="f16 or bf16"):
        _lib.call("vivim_wgrad_nt", P, torch.cuda.current_stream().cuda_stream)
