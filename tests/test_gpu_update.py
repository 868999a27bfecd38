"""GPU: the single-token steps (csrc/update.hip) through the reference's Python names, against the reference's own
fixtures and against the C oracle on random shapes, including chained steps versus the full-sequence kernels."""
import pytest
import torch

from conftest import DT, golden_names, load_golden, rel_err
from oracle import cpu_oracle

pytestmark = pytest.mark.gpu


def _opt(g, k, dev, dtype=None):
    if k not in g:
        return None
    t = g[k].to(dev)
    return t.to(dtype) if dtype is not None else t


@pytest.mark.parametrize("name", golden_names("update_conv_"))
def test_conv_update_golden(name, cuda):
    from causal_conv1d import causal_conv1d_update
    g = load_golden(name)
    dt = DT[g["dtype"]]
    silu = bool(g["meta"][4])
    st = g["conv_state"].to(cuda).to(dt)
    out = causal_conv1d_update(g["x"].to(cuda).to(dt), st, g["weight"].to(cuda), _opt(g, "bias", cuda),
                               "silu" if silu else None)
    assert torch.equal(st.float().cpu(), g["conv_state_new"])             # shifted window: bit-exact
    assert rel_err(out.float(), g["out"]) < (2e-5 if dt == torch.float32 else 1e-2)


@pytest.mark.parametrize("name", golden_names("update_ssm_"))
def test_state_update_golden(name, cuda):
    from mamba_ssm.ops.triton.selective_state_update import selective_state_update
    g = load_golden(name)
    dt = DT[g["dtype"]]
    sp = bool(g["meta"][6])
    st = g["state"].to(cuda)                                               # fp32 state
    out = selective_state_update(st, g["x"].to(cuda).to(dt), g["dt"].to(cuda).to(dt), g["A"].to(cuda),
                                 g["B"].to(cuda).to(dt), g["C"].to(cuda).to(dt), _opt(g, "D", cuda),
                                 _opt(g, "z", cuda, dt), _opt(g, "dt_bias", cuda), sp)
    # the reference ref rounds intermediates to the 16-bit dtype, its kernel (and ours) works in fp32
    assert rel_err(out.float(), g["out"]) < (2e-5 if dt == torch.float32 else 2e-2)
    assert rel_err(st, g["state_new"]) < (2e-5 if dt == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("batch,dim,width", [(1, 1, 2), (3, 300, 4), (2, 1024, 3)])
def test_conv_update_vs_oracle_and_full_conv(dtype, batch, dim, width, cuda):
    """Random cases against the C oracle; then L chained steps from a zero window reproduce causal_conv1d_fn."""
    from causal_conv1d import causal_conv1d_fn, causal_conv1d_update
    g = torch.Generator().manual_seed(dim + width)
    w, b = torch.randn(dim, width, generator=g).to(cuda), torch.randn(dim, generator=g).to(cuda)
    st = torch.randn(batch, dim, width, generator=g).to(dtype).to(cuda)
    x = torch.randn(batch, dim, generator=g).to(dtype).to(cuda)
    r_out, r_st = cpu_oracle.causal_conv1d_update(x, st, w, b, True)
    out = causal_conv1d_update(x, st, w, b, "silu")
    assert torch.equal(st.float().cpu(), r_st)
    assert rel_err(out.float(), r_out.to(dtype).float()) < (1e-5 if dtype == torch.float32 else 8e-3)
    L = 9
    seq = torch.randn(batch, dim, L, generator=g).to(dtype).to(cuda)
    full = causal_conv1d_fn(seq, w, b, "silu")
    st = torch.zeros(batch, dim, width, device=cuda, dtype=dtype)
    for t in range(L):
        step = causal_conv1d_update(seq[:, :, t].contiguous(), st, w, b, "silu")
        assert rel_err(step.float(), full[:, :, t].float()) < (1e-5 if dtype == torch.float32 else 8e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("batch,dim,N", [(1, 3, 1), (2, 128, 16), (3, 70, 64), (1, 16, 200)])
def test_state_update_vs_oracle_and_full_scan(dtype, batch, dim, N, cuda):
    """Random cases against the C oracle (strided inputs too); then L chained steps reproduce selective_scan_fn
    (output and last state)."""
    from mamba_ssm.ops.selective_scan_interface import selective_scan_fn
    from mamba_ssm.ops.triton.selective_state_update import selective_state_update
    g = torch.Generator().manual_seed(dim * 7 + N)
    A = (-0.5 * torch.rand(dim, N, generator=g) - 0.05).to(cuda)
    D, bias = torch.randn(dim, generator=g).to(cuda), (0.5 * torch.rand(dim, generator=g)).to(cuda)
    mk = lambda *s: torch.randn(*s, generator=g).to(dtype).to(cuda)
    st = torch.randn(batch, dim, N, generator=g).to(cuda)
    x, z, dtv = mk(dim, batch).t(), mk(batch, dim), (0.5 * torch.rand(batch, dim, generator=g)).to(dtype).to(cuda)
    Bm, Cm = mk(batch, N), mk(batch, N)
    r_out, r_st = cpu_oracle.selective_state_update(st, x, dtv, A, Bm, Cm, D, z, bias, True)
    out = selective_state_update(st, x, dtv, A, Bm, Cm, D, z, bias, True)
    assert rel_err(st.cpu(), r_st) < 1e-5
    assert rel_err(out.float(), r_out.to(dtype).float()) < (1e-5 if dtype == torch.float32 else 8e-3)
    L = 11
    u, dl, zz = mk(batch, dim, L), (0.5 * torch.rand(batch, dim, L, generator=g)).to(dtype).to(cuda), mk(batch, dim, L)
    Bs, Cs = mk(batch, N, L), mk(batch, N, L)
    full, last = selective_scan_fn(u, dl, A, Bs, Cs, D, zz, bias, True, return_last_state=True)
    st = torch.zeros(batch, dim, N, device=cuda)
    for t in range(L):
        step = selective_state_update(st, u[:, :, t], dl[:, :, t], A, Bs[:, :, t], Cs[:, :, t], D, zz[:, :, t], bias, True)
        assert rel_err(step.float(), full[:, :, t].float()) < (2e-5 if dtype == torch.float32 else 1e-2)
    assert rel_err(st, last) < 2e-5


def test_update_errors(cuda):
    from causal_conv1d import causal_conv1d_update
    from mamba_ssm.ops.triton.selective_state_update import selective_state_update
    x, st, w = torch.randn(2, 8, device=cuda), torch.randn(2, 8, 4, device=cuda), torch.randn(8, 4, device=cuda)
    with pytest.raises(NotImplementedError):
        causal_conv1d_update(x, st, w, None, "gelu")                      # causal_conv1d_interface.py:77-78
    with pytest.raises(RuntimeError):
        causal_conv1d_update(x, st, torch.randn(8, 5, device=cuda), None, None)   # width 5
    with pytest.raises(RuntimeError):
        causal_conv1d_update(x, st.half(), w, None, None)                 # state dtype != x dtype
    with pytest.raises(RuntimeError):
        selective_state_update(torch.randn(2, 8, 4, device=cuda), x, x, torch.randn(8, 5, device=cuda),
                               torch.randn(2, 4, device=cuda), torch.randn(2, 4, device=cuda))


def test_module_step_matches_full_sequence(cuda):
    """Mamba.step (the reference's stock single-direction decode step, mamba_simple.py:356-399) chained over a sequence
    reproduces the full-sequence forward-direction path built from the same parameters (in_proj -> fused inner op ->
    out_proj, i.e. mamba_inner_fn of mamba_simple.py:300-323)."""
    from mamba_ssm import Mamba
    from mamba_ssm.ops.selective_scan_interface import mamba_inner_fn
    torch.manual_seed(3)
    m = Mamba(d_model=32, d_state=16, d_conv=4, expand=2, bimamba_type="v3").to(cuda)
    B, L = 2, 24
    x = torch.randn(B, L, 32, device=cuda)
    with torch.no_grad():
        xz = (m.in_proj.weight @ x.reshape(B * L, -1).t()).view(2 * m.d_inner, B, L).transpose(0, 1)
        full = mamba_inner_fn(xz, m.conv1d.weight, m.conv1d.bias, m.x_proj.weight, m.dt_proj.weight, m.out_proj.weight,
                              m.out_proj.bias, -torch.exp(m.A_log.float()), None, None, m.D.float(),
                              delta_bias=m.dt_proj.bias.float(), delta_softplus=True)
        conv_state, ssm_state = m.allocate_inference_cache(B, L)
        outs = [m.step(x[:, t:t + 1], conv_state, ssm_state)[0] for t in range(L)]
    assert rel_err(torch.cat(outs, dim=1), full) < 2e-5
