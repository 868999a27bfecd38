"""vivim_amd -- MI355X-native (gfx950) implementation of Vivim's Temporal-Mamba-Block hot path:
causal_conv1d fwd/bwd + selective scan fwd/bwd as hand-written HIP kernels behind a C ABI
(include/vivim_hip.h), plus the host-side mirror of the reference's Python surface.

There is no CPU or PyTorch fallback: importing the op modules without the built
vivim_amd/csrc/libvivim_hip.so raises ImportError on first use.
"""
__version__ = "0.1.0"
