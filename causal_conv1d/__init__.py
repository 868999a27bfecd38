"""Reference package name `causal_conv1d` (causal-conv1d/causal_conv1d/__init__.py) -> vivim_amd."""
__version__ = "1.0.0"
from vivim_amd.causal_conv1d_interface import causal_conv1d_fn, causal_conv1d_update  # noqa: F401
