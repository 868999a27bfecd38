"""CPU: the C-ABI shared library loads without a GPU, exports every symbol include/vivim_hip.h declares,
agrees with the ctypes struct layouts, and rejects bad params before any launch."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from vivim_amd import _lib


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "vivim_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vivim_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    L = _lib.lib()
    names = _declared_functions()
    assert set(_lib.EXPORTS) == set(names)
    for n in names:
        assert hasattr(L, n), n


def test_struct_layouts_match():
    L = _lib.lib()
    for which, st in enumerate((_lib.SsmFwdParams, _lib.SsmBwdParams, _lib.ConvFwdParams, _lib.ConvBwdParams,
                                _lib.DwConvParams, _lib.DwConvWgradParams, _lib.DirParams, _lib.ConvUpdateParams,
                                _lib.StateUpdateParams, _lib.LayerNormParams, _lib.WgradNtParams)):
        assert L.vivim_sizeof(which) == ctypes.sizeof(st)
    assert L.vivim_sizeof(99) == 0
    assert L.vivim_abi_version() == 8
    assert L.vivim_scan_chunk_len(_lib.F32) > 0 and L.vivim_scan_chunk_len(_lib.BF16) % 64 == 0


def test_checkpoint_length_is_a_function_of_shape_and_tuning():
    """vivim_scan_ckpt_len: short rows (16 tokens per dstate-16 row) for the shapes the lanes=states backward is chosen for,
    the long rows otherwise; forward tuning 1-3 (families that write long rows) and 5 / 6 (short rows) override the choice."""
    L = _lib.lib()
    s = _lib.SsmFwdParams()
    s.batch, s.dim, s.n_groups, s.dstate, s.seqlen, s.itype = 2, 64, 1, 16, 1280, _lib.BF16
    s.is_variable_B = s.is_variable_C = 1
    s.u_d_stride = s.delta_d_stride = s.B_dstate_stride = s.C_dstate_stride = 1280
    long_rows = L.vivim_scan_chunk_len(_lib.BF16)
    prev = L.vivim_set_tuning(0, 0)
    try:
        assert L.vivim_scan_ckpt_len(ctypes.byref(s)) == 16
        s.seqlen = 20480                                  # 16-bit rows up to 32768 tokens: still the lanes=states backward (round 3)
        s.u_d_stride = s.delta_d_stride = s.B_dstate_stride = s.C_dstate_stride = 20480
        assert L.vivim_scan_ckpt_len(ctypes.byref(s)) == 16
        s.itype = _lib.F32                                # fp32 rows longer than 8192: the lanes=tokens backward, long checkpoint rows
        assert L.vivim_scan_ckpt_len(ctypes.byref(s)) == L.vivim_scan_chunk_len(_lib.F32)
        s.itype = _lib.BF16
        s.seqlen = 40960
        s.u_d_stride = s.delta_d_stride = s.B_dstate_stride = s.C_dstate_stride = 40960
        assert L.vivim_scan_ckpt_len(ctypes.byref(s)) == long_rows
        L.vivim_set_tuning(0, 6)
        assert L.vivim_scan_ckpt_len(ctypes.byref(s)) == 16
        s.dstate = 64
        assert L.vivim_scan_ckpt_len(ctypes.byref(s)) == 64
        s.dstate = 24                                     # not a whole number of 16-lane rows
        assert L.vivim_scan_ckpt_len(ctypes.byref(s)) == long_rows
        s.dstate = 16
        L.vivim_set_tuning(0, 1)
        assert L.vivim_scan_ckpt_len(ctypes.byref(s)) == long_rows
        s.is_variable_B = s.is_variable_C = 0
        L.vivim_set_tuning(0, 0)
        assert L.vivim_scan_ckpt_len(ctypes.byref(s)) == long_rows
    finally:
        L.vivim_set_tuning(0, prev)


def test_rejects_before_launch():
    """Invalid params are refused on the host (no GPU needed): null struct, bad width, bad dtype."""
    L = _lib.lib()
    assert L.vivim_selective_scan_fwd(None, None) == 1
    assert b"check failed" in L.vivim_last_error()
    p = _lib.ConvFwdParams()
    p.batch, p.dim, p.seqlen, p.width = 1, 4, 8, 5
    p.x, p.weight, p.out = 1, 1, 1
    p.x_l_stride = p.out_l_stride = 1
    assert L.vivim_causal_conv1d_fwd(ctypes.byref(p), None) == 1
    assert b"width between 2 and 4" in L.vivim_last_error()
    p.width, p.itype = 4, 7
    assert L.vivim_causal_conv1d_fwd(ctypes.byref(p), None) == 1
    p.itype, p.x_l_stride, p.x_c_stride = 0, 4, 2
    assert L.vivim_causal_conv1d_fwd(ctypes.byref(p), None) == 1          # neither seqlen nor channels contiguous
    assert b"unit stride along seqlen or along channels" in L.vivim_last_error()
    p.x_c_stride = 1                                                      # channel-last x needs a channel-last out
    assert L.vivim_causal_conv1d_fwd(ctypes.byref(p), None) == 1
    s = _lib.SsmFwdParams()
    s.batch = s.dim = s.seqlen = s.n_groups = 1
    s.dstate = 300
    assert L.vivim_selective_scan_fwd(ctypes.byref(s), None) == 1


def test_call_raises_runtime_error():
    with pytest.raises(RuntimeError, match="check failed"):
        _lib.call("vivim_selective_scan_fwd", _lib.SsmFwdParams(), 0)


def test_python_surface_imports_without_gpu():
    import causal_conv1d  # noqa: F401
    import causal_conv1d_cuda
    import mamba_ssm
    import selective_scan_cuda
    from modeling.vivim import MambaLayer, Vivim, mamba_block  # noqa: F401
    assert callable(selective_scan_cuda.fwd) and callable(selective_scan_cuda.bwd)
    assert callable(causal_conv1d_cuda.causal_conv1d_fwd) and callable(causal_conv1d_cuda.causal_conv1d_bwd)
    assert hasattr(mamba_ssm, "Mamba") and hasattr(mamba_ssm, "selective_scan_fn")


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under vivim_amd/ (or the alias packages) may reference it."""
    bad = []
    for top in ("vivim_amd", "mamba_ssm", "causal_conv1d", "modeling"):
        for dp, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".cuh", ".h")):
                    src = open(os.path.join(dp, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b|ssm_oracle|ref_torch", src, flags=re.M):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_tuning_roundtrip():
    """vivim_set_tuning returns the previous value and rejects bad arguments (no GPU needed)."""
    L = _lib.lib()
    prev = L.vivim_set_tuning(0, 5)
    assert prev >= 0
    assert L.vivim_set_tuning(0, prev) == 5
    assert L.vivim_set_tuning(2, 0) == -1 and L.vivim_set_tuning(0, -3) == -1


def test_dir_maps_reject_bad_arguments_on_host():
    """vivim_dir_scatter / vivim_dir_gather validate before launching (no GPU needed for the rejections)."""
    import ctypes
    L = _lib.lib()
    P = _lib.DirParams()
    P.batch, P.channels, P.seqlen, P.nframes, P.csplit, P.itype, P.scale = 1, 4, 30, 5, 2, _lib.BF16, 1.0
    P.src = P.dst = 0
    for fn in (L.vivim_dir_scatter, L.vivim_dir_gather):
        assert fn(ctypes.byref(P), None) != 0                      # null pointers
    P.src = P.dst = 4096
    assert L.vivim_dir_scatter(ctypes.byref(P), None) != 0        # 30 bf16 tokens are not whole 16-byte vectors
    P.seqlen, P.nframes = 32, 5
    assert L.vivim_dir_scatter(ctypes.byref(P), None) != 0        # seqlen not a multiple of nframes
    assert b"" != L.vivim_last_error()


def test_graft_entry_build_is_consistent():
    """__graft_entry__.build() (what the driver runs every round) must accept the library it has just built: its ABI
    assertion is derived from the header, not hard-coded."""
    import __graft_entry__ as g
    g.build()
