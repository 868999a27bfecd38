"""CPU: pins oracle/ (C restatement + torch restatement) against fixtures produced by the reference's
own selective_scan_ref / causal_conv1d_ref / mamba_inner_ref / v3 Mamba.forward (tests/golden)."""
import pytest
import torch

from conftest import DT, golden_names, load_golden, rel_err
from oracle import cpu_oracle, ref_torch

# the refs run in fp32; the C oracle accumulates in fp64 -> agreement at fp32 round-off level
TOL32 = 2e-5
TOL16 = {"bf16": 1.2e-2, "fp16": 2e-3}   # fixtures hold outputs rounded to the 16-bit I/O dtype


def _opt(g, k):
    return g[k] if k in g else None


@pytest.mark.parametrize("name", golden_names("scan_"))
def test_c_scan_matches_reference(name):
    g = load_golden(name)
    sp = bool(g["meta"][10])
    out, out_z, last = cpu_oracle.selective_scan_fwd(g["u"], g["delta"], g["A"], g["B"], g["C"], _opt(g, "D"),
                                                     _opt(g, "z"), _opt(g, "delta_bias"), sp)
    y = out_z if out_z is not None else out
    tol = TOL32 if g["dtype"] == "fp32" else TOL16[g["dtype"]]
    assert rel_err(y, g["out"]) < tol
    assert rel_err(last, g["last_state"]) < TOL32 * 5
    gr = cpu_oracle.selective_scan_bwd(g["u"], g["delta"], g["A"], g["B"], g["C"], _opt(g, "D"), _opt(g, "z"),
                                       _opt(g, "delta_bias"), g["dout"], sp)
    gtol = 1e-4 if g["dtype"] == "fp32" else TOL16[g["dtype"]] * 2
    for k, v in gr.items():
        if v is None:
            assert k not in g
            continue
        assert rel_err(v, g[k]) < gtol, k


@pytest.mark.parametrize("name", golden_names("scan_"))
def test_torch_scan_matches_reference(name):
    g = load_golden(name)
    dt = DT[g["dtype"]]
    cast = lambda t: None if t is None else t.to(dt)
    vB, vC = bool(g["meta"][5]), bool(g["meta"][6])
    out, last = ref_torch.selective_scan_ref(
        cast(g["u"]), cast(g["delta"]), g["A"], cast(g["B"]) if vB else g["B"], cast(g["C"]) if vC else g["C"],
        _opt(g, "D"), cast(_opt(g, "z")), _opt(g, "delta_bias"), bool(g["meta"][10]), return_last_state=True)
    # same fp32 arithmetic as the reference's ref up to op ordering
    assert rel_err(out.float(), g["out"]) < (1e-5 if g["dtype"] == "fp32" else TOL16[g["dtype"]])
    assert rel_err(last, g["last_state"]) < 1e-5


@pytest.mark.parametrize("name", golden_names("conv_"))
def test_conv_matches_reference(name):
    g = load_golden(name)
    silu = bool(g["meta"][4])
    tol = TOL32 if g["dtype"] == "fp32" else TOL16[g["dtype"]]
    out = cpu_oracle.causal_conv1d_fwd(g["x"], g["weight"], _opt(g, "bias"), silu)
    assert rel_err(out, g["out"]) < tol
    dx, dw, db = cpu_oracle.causal_conv1d_bwd(g["x"], g["weight"], _opt(g, "bias"), g["dout"], silu)
    assert rel_err(dx, g["dx"]) < max(tol, 5e-5)
    assert rel_err(dw, g["dweight"]) < max(tol, 5e-5)
    if db is not None:
        assert rel_err(db, g["dbias"]) < max(tol, 5e-5)
    dt = DT[g["dtype"]]
    out_t = ref_torch.causal_conv1d_ref(g["x"].to(dt), g["weight"], _opt(g, "bias"), "silu" if silu else None)
    assert rel_err(out_t.float(), g["out"]) < (1e-6 if g["dtype"] == "fp32" else tol)


def test_conv_rejects_unknown_activation():
    with pytest.raises(NotImplementedError):
        ref_torch.causal_conv1d_ref(torch.zeros(1, 2, 4), torch.zeros(2, 3), None, "relu")


@pytest.mark.parametrize("name", golden_names("inner_"))
def test_inner_op_matches_reference(name):
    g = load_golden(name)
    names = ["conv_w", "conv_b", "x_proj", "dt_proj", "A", "D", "dt_bias"]
    p = {k: g[k].clone().requires_grad_(True) for k in names}
    xz = g["xz"].clone().requires_grad_(True)
    out = ref_torch.mamba_inner_no_out_proj_ref(xz, p["conv_w"], p["conv_b"], p["x_proj"], p["dt_proj"],
                                                p["A"], p["D"], p["dt_bias"], True)
    assert rel_err(out, g["out"]) < 1e-5
    out.backward(g["dout"])
    assert rel_err(xz.grad, g["dxz"]) < 1e-4
    for k in names:
        assert rel_err(p[k].grad, g["d" + k]) < 1e-4, k


BIMAMBA_PARAMS = ["conv_w", "conv_b", "x_proj", "dt_proj", "out_w", "out_b", "A", "A_b", "D", "dt_bias"]


@pytest.mark.parametrize("name", golden_names("bimamba_"))
def test_bimamba_inner_matches_reference(name):
    """The restatement of bimamba_inner_ref (selective_scan_interface.py:673-709) against the reference's own output and
    autograd gradients."""
    g = load_golden(name)
    p = {k: g[k].clone().requires_grad_(True) for k in BIMAMBA_PARAMS}
    xz = g["xz"].clone().requires_grad_(True)
    out = ref_torch.bimamba_inner_ref(xz, p["conv_w"], p["conv_b"], p["x_proj"], p["dt_proj"], p["out_w"], p["out_b"],
                                      p["A"], p["A_b"], p["D"], p["dt_bias"], True)
    assert rel_err(out, g["out"]) < 1e-5
    out.backward(g["dout"])
    assert rel_err(xz.grad, g["dxz"]) < 1e-4
    for k in BIMAMBA_PARAMS:
        assert rel_err(p[k].grad, g["d" + k]) < 1e-4, k


@pytest.mark.parametrize("name", golden_names("module_"))
def test_v3_module_matches_reference(name):
    g = load_golden(name)
    nf = g["meta"][4]
    p = {k[4:].replace("__", "."): v.clone().requires_grad_(True) for k, v in g.items() if k.startswith("sd__")}
    x = g["x"].clone().requires_grad_(True)
    y = ref_torch.mamba_v3_forward_ref(x, p, nf)
    assert rel_err(y, g["y"]) < 1e-5
    y.backward(g["dout"])
    assert rel_err(x.grad, g["dx"]) < 1e-4
    for k, v in p.items():
        assert rel_err(v.grad, g["grad__" + k.replace(".", "__")]) < 2e-4, k


@pytest.mark.parametrize("name", golden_names("update_conv_"))
def test_c_conv_update_matches_reference(name):
    """oracle_causal_conv1d_update against causal_conv1d_update_ref fixtures (causal_conv1d_interface.py:83-104)."""
    g = load_golden(name)
    silu = bool(g["meta"][4])
    out, st = cpu_oracle.causal_conv1d_update(g["x"], g["conv_state"], g["weight"], _opt(g, "bias"), silu)
    tol = TOL32 if g["dtype"] == "fp32" else TOL16[g["dtype"]]
    assert rel_err(out, g["out"]) < tol
    assert torch.equal(st, g["conv_state_new"])            # a pure shift: bit-exact


@pytest.mark.parametrize("name", golden_names("update_ssm_"))
def test_c_state_update_matches_reference(name):
    """oracle_selective_state_update against selective_state_update_ref fixtures (selective_state_update.py:157-192)."""
    g = load_golden(name)
    sp = bool(g["meta"][6])
    out, st = cpu_oracle.selective_state_update(g["state"], g["x"], g["dt"], g["A"], g["B"], g["C"], _opt(g, "D"),
                                                _opt(g, "z"), _opt(g, "dt_bias"), sp)
    tol = TOL32 if g["dtype"] == "fp32" else TOL16[g["dtype"]]
    assert rel_err(out, g["out"]) < tol
    assert rel_err(st, g["state_new"]) < (TOL32 if g["dtype"] == "fp32" else 1e-2)
