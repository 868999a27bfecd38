from vivim_amd.selective_scan_interface import (  # noqa: F401
    BiMambaInnerFn, MambaInnerFnNoOutProj, SelectiveScanFn, bimamba_inner_fn, mamba_inner_fn, mamba_inner_fn_no_out_proj,
    selective_scan_fn)
