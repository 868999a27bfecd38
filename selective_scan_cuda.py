"""`import selective_scan_cuda` resolves here: the reference's extension-module name
(mamba/csrc/selective_scan/selective_scan.cpp:494-497) backed by the gfx950 kernels."""
from vivim_amd.selective_scan_cuda import bwd, fwd, last_workspace_bytes  # noqa: F401
