"""Reference module path mamba_ssm/ops/triton/selective_state_update.py -> the HIP implementation."""
from vivim_amd.selective_state_update import selective_state_update  # noqa: F401
