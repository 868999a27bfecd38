"""Split-k weight-gradient products of the fused inner op's backward (csrc/wgrad.hip; include/vivim_hip.h:
vivim_wgrad_nt_params) -- the two einsum calls of mamba_ssm/ops/selective_scan_interface.py:273, 276.

    wgrad_nt(a, b)      a: (G, M, K), b: (G, N, K), unit stride along K  ->  (G, M, N) float32 = a @ b^T per group

`supported(a, b)` says whether the kernel applies (f16 / bf16 operands, aligned rows) and pays: from 8 192 tokens up (stages 0 and
1 of the 256 x 256 configs: 45 us against 136 and 26 against 30 for the two products); below that the library GEMM has enough
output tiles of its own and is as fast (profiles/r03_wgrad_sweep.txt).  The caller keeps torch.bmm otherwise."""
import torch

from . import _lib

_ITYPE = {torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}
MIN_TOKENS = 8192        # below: torch.bmm (tests set it to 0 to reach the kernel with small problems)


def supported(a, b):
    if not (a.is_cuda and a.dim() == 3 and b.dim() == 3 and a.dtype in _ITYPE and b.dtype == a.dtype
            and a.shape[0] == b.shape[0] and a.shape[2] == b.shape[2]):
        return False
    ok = lambda t: (t.stride(2) == 1 and t.stride(1) % 8 == 0 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0)  # noqa: E731
    return a.shape[2] % 8 == 0 and a.shape[2] >= MIN_TOKENS and ok(a) and ok(b) and a.shape[0] <= 65535


def wgrad_nt(a, b):
    G, M, K = a.shape
    N = b.shape[1]
    out = _lib.zeros(G * M * N, a.device).view(G, M, N)
    P = _lib.WgradNtParams()
    P.groups, P.m, P.n, P.k, P.itype = G, M, N, K, _ITYPE[a.dtype]
    P.a_group_stride, P.a_row_stride = a.stride(0), a.stride(1)
    P.b_group_stride, P.b_row_stride = b.stride(0), b.stride(1)
    P.out_group_stride, P.out_row_stride = M * N, N
    P.a, P.b, P.out = a.data_ptr(), b.data_ptr(), out.data_ptr()
    if a.device.index == torch.cuda.current_device():
        _lib.call("vivim_wgrad_nt", P, torch.cuda.current_stream().cuda_stream)
    else:
        with torch.cuda.device(a.device):
            _lib.call("vivim_wgrad_nt", P, torch.cuda.current_stream().cuda_stream)
    return out
