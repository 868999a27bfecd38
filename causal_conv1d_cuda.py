"""`import causal_conv1d_cuda` resolves here: the reference's extension-module name
(causal-conv1d/csrc/causal_conv1d.cpp:329-333) backed by the gfx950 kernels."""
from vivim_amd.causal_conv1d_cuda import causal_conv1d_bwd, causal_conv1d_fwd, causal_conv1d_update  # noqa: F401
