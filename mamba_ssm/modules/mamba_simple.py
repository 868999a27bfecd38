from vivim_amd.mamba_simple import Mamba  # noqa: F401
