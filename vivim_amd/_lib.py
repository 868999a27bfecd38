"""ctypes binding of libvivim_hip.so (C ABI: include/vivim_hip.h).

The library is the product: if it is missing or fails to load this module raises -- there is no
PyTorch/CPU fallback anywhere in vivim_amd.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(CSRC, "libvivim_hip.so")

i32, i64, vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p

F32, F16, BF16 = 0, 1, 2


class SsmFwdParams(ctypes.Structure):
    _fields_ = (
        [(n, i32) for n in ("batch", "dim", "seqlen", "dstate", "n_groups", "itype", "is_variable_B",
                            "is_variable_C", "delta_softplus", "_pad0")]
        + [(n, i64) for n in ("u_batch_stride", "u_d_stride", "delta_batch_stride", "delta_d_stride",
                              "z_batch_stride", "z_d_stride", "out_batch_stride", "out_d_stride",
                              "out_z_batch_stride", "out_z_d_stride", "A_d_stride", "A_dstate_stride",
                              "B_batch_stride", "B_group_stride", "B_dstate_stride",
                              "C_batch_stride", "C_group_stride", "C_dstate_stride")]
        + [(n, vp) for n in ("u", "delta", "A", "B", "C", "D", "delta_bias", "z", "out", "out_z", "x")]
    )


class SsmBwdParams(ctypes.Structure):
    _fields_ = (
        [("f", SsmFwdParams)]
        + [(n, i64) for n in ("dout_batch_stride", "dout_d_stride", "du_batch_stride", "du_d_stride",
                              "ddelta_batch_stride", "ddelta_d_stride", "dz_batch_stride", "dz_d_stride",
                              "dA_d_stride", "dA_dstate_stride",
                              "dB_batch_stride", "dB_group_stride", "dB_dstate_stride",
                              "dC_batch_stride", "dC_group_stride", "dC_dstate_stride")]
        + [(n, vp) for n in ("dout", "du", "ddelta", "dz", "dA", "dB", "dC", "dD", "ddelta_bias")]
    )


class ConvFwdParams(ctypes.Structure):
    _fields_ = (
        [(n, i32) for n in ("batch", "dim", "seqlen", "width", "itype", "wtype", "silu_activation", "_pad0")]
        + [(n, i64) for n in ("x_batch_stride", "x_c_stride", "x_l_stride",
                              "out_batch_stride", "out_c_stride", "out_l_stride",
                              "weight_c_stride", "weight_width_stride")]
        + [(n, vp) for n in ("x", "weight", "bias", "out")]
    )


class ConvBwdParams(ctypes.Structure):
    _fields_ = (
        [("f", ConvFwdParams)]
        + [(n, i64) for n in ("dout_batch_stride", "dout_c_stride", "dout_l_stride",
                              "dx_batch_stride", "dx_c_stride", "dx_l_stride",
                              "dweight_c_stride", "dweight_width_stride")]
        + [(n, vp) for n in ("dout", "dx", "dweight", "dbias")]
    )


EXPORTS = ("vivim_abi_version", "vivim_last_error", "vivim_scan_chunk_len", "vivim_sizeof",
           "vivim_selective_scan_fwd", "vivim_selective_scan_bwd",
           "vivim_causal_conv1d_fwd", "vivim_causal_conv1d_bwd")

_lib = None


def build(force=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-j4"])
    return SO_PATH


def lib():
    """The loaded library; raises if it is not built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C vivim_amd/csrc`). vivim_amd has no fallback path.")
        L = ctypes.CDLL(SO_PATH)
        L.vivim_last_error.restype = ctypes.c_char_p
        L.vivim_sizeof.restype = ctypes.c_size_t
        L.vivim_sizeof.argtypes = [ctypes.c_int]
        for name, st in (("vivim_selective_scan_fwd", SsmFwdParams), ("vivim_selective_scan_bwd", SsmBwdParams),
                         ("vivim_causal_conv1d_fwd", ConvFwdParams), ("vivim_causal_conv1d_bwd", ConvBwdParams)):
            fn = getattr(L, name)
            fn.argtypes = [ctypes.POINTER(st), vp]
            fn.restype = ctypes.c_int
        if L.vivim_abi_version() != 1:
            raise ImportError("libvivim_hip.so ABI version mismatch")
        for which, st in enumerate((SsmFwdParams, SsmBwdParams, ConvFwdParams, ConvBwdParams)):
            if L.vivim_sizeof(which) != ctypes.sizeof(st):
                raise ImportError(f"struct layout mismatch for {st.__name__}: "
                                  f"C {L.vivim_sizeof(which)} vs ctypes {ctypes.sizeof(st)}")
        _lib = L
    return _lib


def call(name, params, stream):
    """Enqueue one entry point on `stream` (int hipStream_t); RuntimeError on a nonzero return,
    like the TORCH_CHECKs of the reference bindings."""
    L = lib()
    rc = getattr(L, name)(ctypes.byref(params), vp(stream))
    if rc != 0:
        raise RuntimeError(L.vivim_last_error().decode())
