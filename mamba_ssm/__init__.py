"""Reference package name `mamba_ssm` (mamba/mamba_ssm/__init__.py) -> vivim_amd.  The reference's
__init__ also imports its language-model stack (MambaLMHeadModel), which Vivim never uses and which
does not import on current `transformers`; it is not part of this package."""
__version__ = "1.0.1"
from vivim_amd.selective_scan_interface import (  # noqa: F401
    bimamba_inner_fn, mamba_inner_fn, mamba_inner_fn_no_out_proj, selective_scan_fn)
from vivim_amd.mamba_simple import Mamba  # noqa: F401
