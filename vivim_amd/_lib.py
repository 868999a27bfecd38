"""ctypes binding of libvivim_hip.so (C ABI: include/vivim_hip.h).

The library is the product: if it is missing or fails to load this module raises -- there is no
PyTorch/CPU fallback anywhere in vivim_amd.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.environ.get("VIVIM_LIB") or os.path.join(CSRC, "libvivim_hip.so")   # VIVIM_LIB: A/B builds

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
        + [(n, vp) for n in ("u", "delta", "A", "B", "C", "D", "delta_bias", "z", "out", "out_z", "x", "workspace")]
        + [("workspace_bytes", i64)]
    )


class SsmBwdParams(ctypes.Structure):
    _fields_ = (
        [("f", SsmFwdParams)]
        + [(n, i64) for n in ("dout_batch_stride", "dout_d_stride", "du_batch_stride", "du_d_stride",
                              "ddelta_batch_stride", "ddelta_d_stride", "dz_batch_stride", "dz_d_stride",
                              "dA_d_stride", "dA_dstate_stride",
                              "dB_batch_stride", "dB_group_stride", "dB_dstate_stride",
                              "dC_batch_stride", "dC_group_stride", "dC_dstate_stride")]
        + [(n, vp) for n in ("dout", "du", "ddelta", "dz", "dA", "dB", "dC", "dD", "ddelta_bias", "workspace")]
        + [("workspace_bytes", i64)]
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


class DwConvParams(ctypes.Structure):
    _fields_ = (
        [(n, i32) for n in ("batch", "depth", "height", "width", "channels", "kd", "itype", "flip")]
        + [(n, i64) for n in ("x_batch_stride", "x_token_stride", "y_batch_stride", "y_token_stride")]
        + [(n, vp) for n in ("x", "wt", "bias", "y")]
        + [("act", i32), ("_pad1", i32), ("aux", vp), ("aux_batch_stride", i64), ("aux_token_stride", i64)]
    )


class DwConvWgradParams(ctypes.Structure):
    _fields_ = (
        [(n, i32) for n in ("batch", "depth", "height", "width", "channels", "kd", "itype", "_pad0")]
        + [(n, i64) for n in ("x_batch_stride", "x_token_stride", "dy_batch_stride", "dy_token_stride")]
        + [(n, vp) for n in ("x", "dy", "dwt", "dbias")]
    )


class DirParams(ctypes.Structure):
    _fields_ = ([(n, i32) for n in ("batch", "channels", "seqlen", "nframes", "csplit", "itype")]
                + [("scale", ctypes.c_float), ("_pad0", i32)]
                + [(n, i64) for n in ("flat_batch_stride", "flat_c_stride", "stk_batch_stride", "stk_half_stride",
                                      "stk_dir_stride", "stk_c_stride")]
                + [("src", vp), ("dst", vp)])


class ConvUpdateParams(ctypes.Structure):
    _fields_ = ([(n, i32) for n in ("batch", "dim", "width", "itype", "wtype", "silu_activation")]
                + [(n, i64) for n in ("x_batch_stride", "x_c_stride", "state_batch_stride", "state_c_stride",
                                      "state_w_stride", "weight_c_stride", "weight_width_stride",
                                      "out_batch_stride", "out_c_stride")]
                + [(n, vp) for n in ("x", "conv_state", "weight", "bias", "out")])


class StateUpdateParams(ctypes.Structure):
    _fields_ = ([(n, i32) for n in ("batch", "dim", "dstate", "itype", "stype", "dt_softplus")]
                + [(n, i64) for n in ("state_batch_stride", "state_d_stride", "state_n_stride", "x_batch_stride",
                                      "x_d_stride", "dt_batch_stride", "dt_d_stride", "A_d_stride", "A_n_stride",
                                      "B_batch_stride", "B_n_stride", "C_batch_stride", "C_n_stride",
                                      "z_batch_stride", "z_d_stride", "out_batch_stride", "out_d_stride")]
                + [(n, vp) for n in ("state", "x", "dt", "A", "B", "C", "D", "z", "dt_bias", "out")])


class LayerNormParams(ctypes.Structure):
    _fields_ = ([(n, i32) for n in ("batch", "seqlen", "channels", "itype", "otype")] + [("eps", ctypes.c_float)]
                + [(n, i64) for n in ("x_batch_stride", "x_c_stride", "y_batch_stride", "y_token_stride",
                                      "dx_batch_stride", "dx_c_stride")]
                + [(n, vp) for n in ("x", "weight", "bias", "y", "mean", "rstd", "dy", "dx", "dweight", "dbias", "workspace")])


class WgradNtParams(ctypes.Structure):
    _fields_ = ([(n, i32) for n in ("groups", "m", "n", "k", "itype", "_pad0")]
                + [(n, i64) for n in ("a_group_stride", "a_row_stride", "b_group_stride", "b_row_stride",
                                      "out_group_stride", "out_row_stride")]
                + [(n, vp) for n in ("a", "b", "out")])


EXPORTS = ("vivim_abi_version", "vivim_last_error", "vivim_scan_chunk_len", "vivim_scan_ckpt_len", "vivim_sizeof",
           "vivim_scan_bwd_workspace_bytes", "vivim_scan_fwd_workspace_bytes", "vivim_set_tuning",
           "vivim_selective_scan_fwd", "vivim_selective_scan_bwd",
           "vivim_causal_conv1d_fwd", "vivim_causal_conv1d_bwd", "vivim_dwconv_fwd", "vivim_dwconv_wgrad",
           "vivim_dir_scatter", "vivim_dir_gather", "vivim_causal_conv1d_update", "vivim_selective_state_update",
           "vivim_layernorm_cm_fwd", "vivim_layernorm_cm_bwd", "vivim_layernorm_bwd_workspace_bytes", "vivim_wgrad_nt")

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
        # torch first: its bundled HIP runtime must be the one this process initialises -- loading the library (and with it
        # /opt/rocm's libamdhip64) before `import torch` leaves two runtimes in the process and every launch of ours then fails
        # with "no ROCm-capable device is detected" (seen with build() followed by smoke() in one process)
        import torch  # noqa: F401
        L = ctypes.CDLL(SO_PATH)
        L.vivim_last_error.restype = ctypes.c_char_p
        L.vivim_sizeof.restype = ctypes.c_size_t
        L.vivim_sizeof.argtypes = [ctypes.c_int]
        L.vivim_set_tuning.restype = ctypes.c_int
        L.vivim_set_tuning.argtypes = [ctypes.c_int, ctypes.c_int]
        L.vivim_scan_ckpt_len.restype = ctypes.c_int
        L.vivim_scan_ckpt_len.argtypes = [ctypes.POINTER(SsmFwdParams)]
        for fn in (L.vivim_scan_bwd_workspace_bytes, L.vivim_scan_fwd_workspace_bytes):
            fn.restype = ctypes.c_size_t
            fn.argtypes = [ctypes.POINTER(SsmFwdParams)]
        for name, st in (("vivim_selective_scan_fwd", SsmFwdParams), ("vivim_selective_scan_bwd", SsmBwdParams),
                         ("vivim_causal_conv1d_fwd", ConvFwdParams), ("vivim_causal_conv1d_bwd", ConvBwdParams),
                         ("vivim_dwconv_fwd", DwConvParams), ("vivim_dwconv_wgrad", DwConvWgradParams),
                         ("vivim_dir_scatter", DirParams), ("vivim_dir_gather", DirParams),
                         ("vivim_causal_conv1d_update", ConvUpdateParams),
                         ("vivim_selective_state_update", StateUpdateParams),
                         ("vivim_layernorm_cm_fwd", LayerNormParams), ("vivim_layernorm_cm_bwd", LayerNormParams),
                         ("vivim_wgrad_nt", WgradNtParams)):
            fn = getattr(L, name)
            fn.argtypes = [ctypes.POINTER(st), vp]
            fn.restype = ctypes.c_int
        L.vivim_layernorm_bwd_workspace_bytes.argtypes = [ctypes.POINTER(LayerNormParams)]
        L.vivim_layernorm_bwd_workspace_bytes.restype = ctypes.c_size_t
        if L.vivim_abi_version() != 8:
            raise ImportError("libvivim_hip.so ABI version mismatch")
        for which, st in enumerate((SsmFwdParams, SsmBwdParams, ConvFwdParams, ConvBwdParams, DwConvParams,
                                    DwConvWgradParams, DirParams, ConvUpdateParams, StateUpdateParams, LayerNormParams,
                                    WgradNtParams)):
            if L.vivim_sizeof(which) != ctypes.sizeof(st):
                raise ImportError(f"struct layout mismatch for {st.__name__}: "
                                  f"C {L.vivim_sizeof(which)} vs ctypes {ctypes.sizeof(st)}")
        _lib = L
    return _lib


_ISIZE = {F32: 4, F16: 2, BF16: 2}


def algorithmic_bytes(name, P):
    """Compulsory HBM traffic of one launch: every tensor of the op read or written once
    (the checkpoint tensor x is an implementation choice and is excluded)."""
    if name.startswith("vivim_selective_scan"):
        f = P.f if name.endswith("bwd") else P
        s = _ISIZE[f.itype]
        act = f.batch * f.dim * f.seqlen * s
        bc = (f.batch * f.n_groups * f.dstate * f.seqlen) if f.is_variable_B else f.dim * f.dstate
        has_z = bool(f.z)
        if name.endswith("fwd"):
            n_act = 3 + (2 if has_z else 0)                       # u, delta, out (+ z, out_z)
            return n_act * act + 2 * bc * (s if f.is_variable_B else 4) + 4 * (f.dim * f.dstate + 2 * f.dim)
        n_act = 5 + (3 if has_z else 0) + (1 if (has_z and f.out_z) else 0)   # u, delta, dout, du, ddelta (+ z, out, dz) (+ out_z)
        return (n_act * act + 2 * bc * (s if f.is_variable_B else 4) + 2 * bc * 4
                + 4 * (2 * f.dim * f.dstate + 4 * f.dim))
    if name.endswith("_update"):
        return 0                                                       # latency-bound single-token steps: not profiled
    if name.startswith("vivim_dir"):
        return 4 * P.batch * P.channels * P.seqlen * _ISIZE[P.itype]  # one flat tensor + three stacked copies
    if name.startswith("vivim_dwconv"):
        act = P.batch * P.depth * P.height * P.width * P.channels * _ISIZE[P.itype]
        return 2 * act + 4 * P.channels * (P.kd * 9 + 1)            # x and y (or x and dy) once + taps
    f = P.f if name.endswith("bwd") else P
    s = _ISIZE[f.itype]
    act = f.batch * f.dim * f.seqlen * s
    if name.endswith("fwd"):
        return 2 * act + 4 * f.dim * (f.width + 1)
    return 3 * act + 8 * f.dim * (f.width + 1)


_profile = None
_event_pool = []
_PROFILED = ("vivim_selective_scan_fwd", "vivim_selective_scan_bwd", "vivim_causal_conv1d_fwd",
             "vivim_causal_conv1d_bwd")


def profile_begin(all_kernels=False):
    """Start recording one (name, algorithmic bytes, start event, end event) tuple per C-ABI call of the hot-path
    kernels (every kernel with all_kernels=True); events are recorded on the stream the kernel is launched on
    (torch's current stream) and come from a reusable pool, so the timed region pays two hipEventRecord per call."""
    global _profile
    _profile = ([], bool(all_kernels))


def profile_end():
    """Stop recording; -> list of (name, bytes, seconds) after synchronising the recorded events."""
    global _profile
    rec = _profile[0] if _profile else []
    _profile = None
    out = []
    for name, nbytes, e0, e1 in rec:
        e1.synchronize()
        out.append((name, nbytes, e0.elapsed_time(e1) * 1e-3))
        _event_pool.extend((e0, e1))
    return out


def _event():
    if _event_pool:
        return _event_pool.pop()
    import torch
    return torch.cuda.Event(enable_timing=True)


# ---- VIVIM_GUARD=1: debug mode that brackets every buffer the wrappers allocate for a kernel to write with canary
# bytes and verifies them (device sync) after each launch -- finds out-of-bounds writes without waiting for a fault.
GUARD = os.environ.get("VIVIM_GUARD", "0") not in ("0", "")
_GUARD_BYTES = 1 << 16
_guards = []


def _guarded_storage(nbytes, device):
    import torch
    buf = torch.full((nbytes + 2 * _GUARD_BYTES,), 0xA5, dtype=torch.uint8, device=device)
    _guards.append(buf)
    return buf[_GUARD_BYTES:_GUARD_BYTES + nbytes]


def empty(shape, dtype, device):
    """torch.empty, or its canary-bracketed twin under VIVIM_GUARD=1."""
    import torch
    if not GUARD:
        return torch.empty(shape, dtype=dtype, device=device)
    n = 1
    for d in shape:
        n *= d
    return _guarded_storage(n * torch.empty((), dtype=dtype).element_size(), device).view(dtype).view(shape)


def zeros(n, device):
    """A zeroed fp32 accumulator of n elements (dA / dB / dC / dD / dbias, dweight / dbias of the convolutions: what the
    kernels ADD into), or its canary-bracketed twin under VIVIM_GUARD=1."""
    import torch
    if not GUARD:
        return torch.zeros(n, dtype=torch.float32, device=device)
    return _guarded_storage(4 * n, device).view(torch.float32).zero_()


def empty_like(t):
    """torch.empty_like (dense, same strides), or its canary-bracketed twin under VIVIM_GUARD=1."""
    import torch
    if not GUARD:
        return torch.empty_like(t)
    ref = torch.empty_like(t)                               # for its strides
    flat = _guarded_storage(t.numel() * t.element_size(), t.device).view(t.dtype)
    return flat.as_strided(ref.shape, ref.stride())


def check_guards(what):
    import torch
    torch.cuda.synchronize()
    for buf in _guards:
        lo, hi = buf[:_GUARD_BYTES], buf[-_GUARD_BYTES:]
        if not (bool((lo == 0xA5).all()) and bool((hi == 0xA5).all())):
            nlo, nhi = int((lo != 0xA5).sum()), int((hi != 0xA5).sum())
            _guards.clear()
            raise RuntimeError(f"VIVIM_GUARD: {what} wrote outside a {buf.numel() - 2 * _GUARD_BYTES}-byte buffer "
                               f"({nlo} bytes below, {nhi} bytes above)")
    _guards.clear()


def call(name, params, stream):
    """Enqueue one entry point on `stream` (int hipStream_t); RuntimeError on a nonzero return,
    like the TORCH_CHECKs of the reference bindings."""
    L = lib()
    if _profile is not None and (_profile[1] or name in _PROFILED):
        e0, e1 = _event(), _event()
        e0.record()
        rc = getattr(L, name)(ctypes.byref(params), vp(stream))
        e1.record()
        _profile[0].append((name, algorithmic_bytes(name, params), e0, e1))
    else:
        rc = getattr(L, name)(ctypes.byref(params), vp(stream))
    if rc != 0:
        raise RuntimeError(L.vivim_last_error().decode())
    if GUARD:
        check_guards(name)
