from vivim_amd.causal_conv1d_interface import CausalConv1dFn, causal_conv1d_fn, causal_conv1d_update  # noqa: F401
