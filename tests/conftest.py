import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
DT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """-> dict of torch tensors (float32) + 'meta' (list of ints) + 'dtype' (str or None)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        if k == "meta":
            out[k] = [int(v) for v in z[k]]
        elif k == "dtype":
            out[k] = str(z[k])
        else:
            out[k] = torch.from_numpy(z[k])
    out.setdefault("dtype", "fp32")
    return out


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


# The reference's own elementwise tolerances (rtol, atol) per I/O dtype: selective scan
# (mamba/tests/ops/test_selective_scan.py:45-51) and causal conv1d (causal-conv1d/tests/test_causal_conv1d.py:31-34).
# The norm-wise bound above is the north_star's 1e-3; a localised defect (one wrong tail token, one channel of a pair, a
# wrong row after a segment cut) barely moves a Frobenius norm over a (B, D, L) tensor, so every parity check also has to
# pass the elementwise comparison the reference itself uses.
SCAN_CLOSE = {torch.float32: (6e-4, 2e-3), torch.float16: (3e-3, 5e-3), torch.bfloat16: (3e-2, 5e-2)}
CONV_CLOSE = {torch.float32: (3e-4, 1e-3), torch.float16: (3e-3, 5e-3), torch.bfloat16: (1e-2, 5e-2)}

_PARITY_LOG = os.path.join(ROOT, "gpurun_out", "parity_relerr.log")


def check_close(name, got, want, dtype, table, norm_tol=None, test=""):
    """Norm-wise rel-err (recorded in gpurun_out/parity_relerr.log, asserted when norm_tol is given) AND the reference's
    elementwise rtol / atol for `dtype`.  `want` is compared as given (round it to the I/O dtype first where the kernel
    rounds)."""
    e = rel_err(got.float(), want.float())
    g, w = got.detach().float().cpu(), want.detach().float().cpu()
    mx = float((g - w).abs().max()) if g.numel() else 0.0
    try:
        os.makedirs(os.path.dirname(_PARITY_LOG), exist_ok=True)
        with open(_PARITY_LOG, "a") as f:
            f.write(f"{test or os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0]}\t{name}\t{str(dtype).replace('torch.', '')}"
                    f"\tshape={tuple(g.shape)}\trel_err={e:.3e}\tmax_abs={mx:.3e}\n")
    except OSError:
        pass
    if norm_tol is not None:
        assert e < norm_tol, f"{name}: rel-err {e:.3e} >= {norm_tol:.1e}"
    rtol, atol = table[dtype]
    torch.testing.assert_close(g, w, rtol=rtol, atol=atol, msg=lambda m: f"{name} ({dtype}): {m}")
    return e


@pytest.fixture(autouse=True)
def _poisoned_allocator(request):
    """GPU tests start with the caching allocator's free blocks full of NaN / 3e38: a kernel or wrapper that reads memory
    nobody wrote (an `empty` buffer whose columns are only partly filled, a halo before a row's first token) then fails its
    parity check instead of passing whenever the allocator happens to hand out zeroed pages.  (It found the unwritten B / C
    columns of dx_dbl in the fused op's constant-B/C backward -- a pattern inherited from the reference,
    selective_scan_interface.py:262-271.)"""
    if request.node.get_closest_marker("gpu") is not None and torch.cuda.is_available():
        fill = float("nan") if (hash(request.node.name) & 1) else 3e38
        junk = [torch.full((1 << k,), fill, device="cuda:0") for k in range(7, 25) for _ in range(2)]
        del junk
    yield


@pytest.fixture(scope="session")
def cuda():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
