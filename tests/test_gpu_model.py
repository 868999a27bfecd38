"""GPU parity of the layers above the kernels: the fused MambaInnerFnNoOutProj op, the v3 Mamba module
and MambaLayer / Vivim, against fixtures produced by the reference (tests/golden, make_golden.py)."""
import pytest
import torch

from conftest import SCAN_CLOSE, check_close, golden_names, load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", golden_names("inner_"))
def test_fused_inner_op_golden(name, cuda):
    """mamba_inner_fn_no_out_proj fwd + all grads vs the reference's mamba_inner_ref with identity out_proj
    (selective_scan_interface.py:155-289 vs :636-670)."""
    from mamba_ssm.ops.selective_scan_interface import mamba_inner_fn_no_out_proj
    g = load_golden(name)
    names = ["conv_w", "conv_b", "x_proj", "dt_proj", "A", "D", "dt_bias"]
    p = {k: g[k].to(cuda).requires_grad_(True) for k in names}
    b, d_inner, n, r, L = g["meta"]
    # hand it the (L, B*L, 1)-strided xz the module produces
    xz = g["xz"].transpose(0, 1).contiguous().to(cuda).transpose(0, 1).requires_grad_(True)
    out = mamba_inner_fn_no_out_proj(xz, p["conv_w"], p["conv_b"], p["x_proj"], p["dt_proj"], p["A"], None, None,
                                     p["D"], delta_bias=p["dt_bias"], delta_softplus=True)
    assert out.shape == (b, d_inner, L)
    check_close("fused.out", out, g["out"], torch.float32, SCAN_CLOSE, 2e-5)
    out.backward(g["dout"].to(cuda))
    check_close("fused.dxz", xz.grad, g["dxz"], torch.float32, SCAN_CLOSE, 2e-4)
    for k in names:
        assert rel_err(p[k].grad, g["d" + k]) < 2e-4, k


@pytest.mark.parametrize("name", golden_names("bimamba_"))
def test_bimamba_inner_fn_golden(name, cuda):
    """bimamba_inner_fn (selective_scan_interface.py:437-603, 616-625) forward + every gradient vs the reference's
    bimamba_inner_ref (:673-709).  (The reference's own test of it compares the function with itself,
    tests/ops/test_selective_scan.py:314-320.)"""
    from mamba_ssm import bimamba_inner_fn
    g = load_golden(name)
    names = ["conv_w", "conv_b", "x_proj", "dt_proj", "out_w", "out_b", "A", "A_b", "D", "dt_bias"]
    p = {k: g[k].to(cuda).requires_grad_(True) for k in names}
    b, d_inner, n, r, L, d_model = g["meta"]
    xz = g["xz"].to(cuda).requires_grad_(True)
    out = bimamba_inner_fn(xz, p["conv_w"], p["conv_b"], p["x_proj"], p["dt_proj"], p["out_w"], p["out_b"],
                           p["A"], p["A_b"], None, None, p["D"], delta_bias=p["dt_bias"], delta_softplus=True)
    assert out.shape == (b, L, d_model)
    check_close("bimamba.out", out, g["out"], torch.float32, SCAN_CLOSE, 2e-5)
    out.backward(g["dout"].to(cuda))
    check_close("bimamba.dxz", xz.grad, g["dxz"], torch.float32, SCAN_CLOSE, 2e-4)
    for k in names:
        assert rel_err(p[k].grad, g["d" + k]) < 2e-4, k


def test_bimamba_inner_fn_constant_BC_matches_two_scans(cuda):
    """Constant (dim, dstate) B and C (the other branch of :474-492) against the composition the reference's ref spells out:
    two selective_scan_fn calls, the second on flipped inputs."""
    from mamba_ssm import bimamba_inner_fn, selective_scan_fn
    from causal_conv1d import causal_conv1d_fn
    gen = torch.Generator().manual_seed(5)
    b, d, n, r, L, e = 2, 24, 16, 2, 96, 10
    mk = lambda *s, k=1.0: (torch.randn(*s, generator=gen) * k).to(cuda).requires_grad_(True)
    xz, cw, cb = mk(b, 2 * d, L), mk(d, 1, 4, k=0.3), mk(d, k=0.1)
    xp, dp, ow, ob = mk(r + 2 * n, d, k=d ** -0.5), mk(d, r, k=r ** -0.5), mk(e, d, k=d ** -0.5), mk(e, k=0.1)
    A = (-torch.rand(d, n, generator=gen) - 0.1).to(cuda).requires_grad_(True)
    A_b = (-torch.rand(d, n, generator=gen) - 0.1).to(cuda).requires_grad_(True)
    Bc, Cc, D, bias = mk(d, n), mk(d, n), mk(d), mk(d, k=0.2)
    leaves = [xz, cw, cb, xp, dp, ow, ob, A, A_b, Bc, Cc, D, bias]
    dout = torch.randn(b, L, e, generator=gen).to(cuda)

    def composed():
        x, z = xz.chunk(2, dim=1)
        x = causal_conv1d_fn(x, cw.squeeze(1), cb, "silu")
        x_dbl = torch.nn.functional.linear(x.transpose(1, 2).reshape(b * L, d), xp)
        delta = (dp @ x_dbl[:, :r].t()).view(d, b, L).transpose(0, 1).contiguous()
        y = selective_scan_fn(x, delta, A, Bc, Cc, D, z=z, delta_bias=bias, delta_softplus=True)
        y_b = selective_scan_fn(x.flip([-1]), delta.flip([-1]), A_b, Bc, Cc, D, z=z.flip([-1]), delta_bias=bias,
                                delta_softplus=True)
        return torch.nn.functional.linear((y + y_b.flip([-1])).transpose(1, 2), ow, ob)

    def grads(fn):
        for t in leaves:
            t.grad = None
        out = fn()
        out.backward(dout)
        return out.detach(), [t.grad.clone() for t in leaves]
    o1, g1 = grads(lambda: bimamba_inner_fn(xz, cw, cb, xp, dp, ow, ob, A, A_b, Bc, Cc, D, delta_bias=bias,
                                            delta_softplus=True))
    o2, g2 = grads(composed)
    assert rel_err(o1, o2) < 2e-5
    for i, (a, c) in enumerate(zip(g1, g2)):
        assert rel_err(a, c) < 2e-4, i


@pytest.mark.parametrize("name", golden_names("module_"))
def test_v3_module_golden(name, cuda):
    """Mamba(bimamba_type='v3') forward/backward vs the reference module run on the reference refs
    (mamba_simple.py:188-264), loading the reference's state dict by name."""
    from mamba_ssm import Mamba
    g = load_golden(name)
    b, d_model, n, expand, nf, hw = g["meta"]
    m = Mamba(d_model=d_model, d_state=n, d_conv=4, expand=expand, bimamba_type="v3", nframes=nf)
    sd = {k[4:].replace("__", "."): v for k, v in g.items() if k.startswith("sd__")}
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    m = m.to(cuda)
    x = g["x"].to(cuda).requires_grad_(True)
    y = m(x)
    check_close("module.y", y, g["y"], torch.float32, SCAN_CLOSE, 2e-5)
    y.backward(g["dout"].to(cuda))
    check_close("module.dx", x.grad, g["dx"], torch.float32, SCAN_CLOSE, 2e-4)
    for k, v in m.named_parameters():
        assert rel_err(v.grad, g["grad__" + k.replace(".", "__")]) < 5e-4, k


def test_fused_parameter_storage(cuda, monkeypatch):
    """The grouped path reads the three directions' parameters from one persistent (3, ...) buffer per kind, of which the
    Parameters are views (mamba_simple.Mamba._fuse): no per-step cat / stack.  Names and shapes are the reference's
    (mamba_simple.py:69-123), gradients reach every Parameter and equal those of the three-call composition, an in-place
    optimizer update is seen by the next forward, and a storage swap (`.half().float()`) is detected and re-fused."""
    from mamba_ssm import Mamba
    torch.manual_seed(3)
    m = Mamba(d_model=32, d_state=16, bimamba_type="v3", nframes=3).to(cuda)
    names = sorted(n for n, _ in m.named_parameters())
    x = torch.randn(2, 3 * 16, 32, device=cuda, requires_grad=True)

    def run():
        for p in m.parameters():
            p.grad = None
        y = m(x)
        y.square().mean().backward()
        return y.detach(), {n: p.grad.clone() for n, p in m.named_parameters()}
    y1, g1 = run()
    assert sorted(n for n, _ in m.named_parameters()) == names and set(g1) == set(names)
    buf = m._fused[2]
    assert buf.shape == (3,) + tuple(m.x_proj.weight.shape)
    assert [m.x_proj.weight.data_ptr(), m.x_proj_b.weight.data_ptr(), m.x_proj_s.weight.data_ptr()] == \
        [buf[g].data_ptr() for g in range(3)]
    monkeypatch.setenv("VIVIM_SEPARATE_DIRECTIONS", "1")
    y2, g2 = run()
    monkeypatch.delenv("VIVIM_SEPARATE_DIRECTIONS")
    assert rel_err(y1, y2) < 2e-5
    for n in names:
        assert rel_err(g1[n], g2[n]) < 2e-4, n
    with torch.no_grad():                                   # an optimizer step in place
        for p in m.parameters():
            p.add_(0.05 * torch.randn_like(p))
    assert m._fused[2].data_ptr() == m.x_proj.weight.data_ptr()
    y3, _ = run()
    monkeypatch.setenv("VIVIM_SEPARATE_DIRECTIONS", "1")
    y4, _ = run()
    monkeypatch.delenv("VIVIM_SEPARATE_DIRECTIONS")
    assert rel_err(y3, y4) < 2e-5 and rel_err(y3, y1) > 1e-3
    m.half().float()                                        # new storage for every Parameter
    assert m._fused[2].data_ptr() != m.x_proj.weight.data_ptr()
    y5, _ = run()
    assert m._fused[2].data_ptr() == m.x_proj.weight.data_ptr()
    assert rel_err(y5, y3) < 2e-3                           # the fp16 round trip of the weights
    # ONE middle Parameter gets new storage / is re-assigned: every fused Parameter is re-checked per call
    m.A_b_log.data = m.A_b_log.data.clone() + 0.3
    m.dt_proj_b.bias = torch.nn.Parameter(m.dt_proj_b.bias.detach().clone() - 0.5)
    assert m._fused[5][1].data_ptr() != m.A_b_log.data_ptr()
    y6, g6 = run()
    assert m._fused[5][1].data_ptr() == m.A_b_log.data_ptr() and m._fused[4][1].data_ptr() == m.dt_proj_b.bias.data_ptr()
    monkeypatch.setenv("VIVIM_SEPARATE_DIRECTIONS", "1")
    y7, g7 = run()
    monkeypatch.delenv("VIVIM_SEPARATE_DIRECTIONS")
    assert rel_err(y6, y7) < 2e-5 and rel_err(y6, y5) > 1e-3
    assert rel_err(g6["dt_proj_b.bias"], g7["dt_proj_b.bias"]) < 2e-4


def test_module_nframes_override_and_autocast(cuda):
    """clip_length != 5 works (reference hard-codes 5, mamba_simple.py:54) and bf16 autocast runs the
    bf16 kernels with fp32 parameters (custom_fwd contract, selective_scan_interface.py:158-171)."""
    from mamba_ssm import Mamba
    torch.manual_seed(0)
    m = Mamba(d_model=64, bimamba_type="v3").to(cuda)
    x = torch.randn(2, 3 * 8 * 8, 64, device=cuda, requires_grad=True)
    y32 = m(x, nframes=3)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y16 = m(x, nframes=3)
    assert y16.dtype == torch.bfloat16
    assert rel_err(y16.float(), y32) < 3e-2
    y16.float().pow(2).mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    with pytest.raises(ValueError):
        m(x, nframes=5)


def test_mamba_layer_and_vivim_train_step(cuda):
    """MambaLayer / Vivim keep the reference's signatures (vivim.py:111-159, 234-348) and a bf16 train
    step yields finite loss and gradients.  Backbone: SegFormer-b3 architecture, random init (no hub)."""
    from modeling.vivim import MambaLayer, Vivim, segformer_b3_random
    torch.manual_seed(0)
    layer = MambaLayer(dim=64).to(cuda)
    x = torch.randn(1, 64, 3, 8, 8, device=cuda)
    assert layer(x).shape == x.shape
    model = Vivim(in_chans=3, out_chans=3, backbone=segformer_b3_random()).to(cuda).train()
    clip = torch.randn(1, 5, 3, 64, 64, device=cuda)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = model(clip)
    assert logits.shape == (5, 3, 64, 64)
    loss = torch.nn.functional.cross_entropy(logits.float(), torch.randint(0, 3, (5, 64, 64), device=cuda))
    loss.backward()
    assert torch.isfinite(loss)
    grads = [p.grad for n, p in model.named_parameters() if "mamba" in n]
    assert grads and all(g is not None and torch.isfinite(g).all() for g in grads)


def test_vivim_inference_matches_training_graph(cuda):
    """no_grad / eval forward at the benchmark's geometry (256x256, clip 5, batch 3): the inference path must run
    and give the same logits as the autograd-recording forward in eval mode (it once died with a GPU memory fault in
    a library GEMM of the decode head; vivim.py:decode keeps the reference's nn.Conv2d there)."""
    from vivim_amd.train_step import build_model, synthetic_batch
    torch.manual_seed(0)
    model = build_model(3, cuda, mamba_kwargs={"d_state": 16, "expand": 2}).eval()
    clip, _ = synthetic_batch(3, 5, 256, 3, cuda, 7)
    torch.manual_seed(1)                        # decode() draws CPU coin flips even in eval (vivim.py:310-312)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        a = model(clip)
    torch.manual_seed(1)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        b = model(clip)
    torch.cuda.synchronize()
    assert a.shape == (15, 3, 256, 256) and torch.isfinite(a).all()
    assert (a.float() - b.float()).abs().max() <= 2e-2 * b.float().abs().max()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_grouped_directions_match_separate_calls(dtype, cuda, monkeypatch):
    """The v3 block with its three directions batched on the channel axis (one conv1d + one scan launch with
    n_groups = 3, MambaInnerGroupedFnNoOutProj) against the reference's call pattern of three separate fused ops:
    same output and same gradients for the input and every parameter."""
    from mamba_ssm import Mamba
    torch.manual_seed(5)
    m = Mamba(d_model=64, d_state=16, d_conv=4, expand=2, bimamba_type="v3", nframes=5).to(cuda)
    x = torch.randn(2, 5 * 48, 64, device=cuda)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("VIVIM_SEPARATE_DIRECTIONS", mode)
        m.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=dtype, enabled=dtype != torch.float32):
            y = m(xi)
        y.float().square().mean().backward()
        res[mode] = (y.detach().float(), xi.grad.float(), {n: p.grad.float().clone() for n, p in m.named_parameters()})
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-20))
    assert rel(res["0"][0], res["1"][0]) < tol
    assert rel(res["0"][1], res["1"][1]) < tol
    for n in res["1"][2]:
        assert rel(res["0"][2][n], res["1"][2][n]) < tol, n


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_grouped_backward_weight_gradients_split_k_kernel(dtype, cuda, monkeypatch):
    """From 8 192 tokens up the grouped op's backward takes csrc/wgrad.hip for x_proj's and dt_proj's weight gradients
    (selective_scan_interface.py:273, 276) instead of torch.bmm: same gradients (the kernel keeps f32 where bmm rounds its
    output to the 16-bit type), everything else unchanged."""
    from mamba_ssm import Mamba
    from vivim_amd import wgrad
    torch.manual_seed(9)
    m = Mamba(d_model=64, d_state=16, d_conv=4, expand=2, bimamba_type="v3", nframes=4).to(cuda)
    x = torch.randn(2, 4 * 1024, 64, device=cuda)                    # 8 192 tokens
    calls = []
    real = wgrad.wgrad_nt
    monkeypatch.setattr(wgrad, "wgrad_nt", lambda a, b: (calls.append(a.shape), real(a, b))[1])
    res = {}
    for mode in ("", "1"):
        if mode:
            monkeypatch.setenv("VIVIM_NO_WGRAD_KERNEL", mode)
        m.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=dtype):
            y = m(xi)
        y.float().square().mean().backward()
        res[mode] = (xi.grad.clone(), {n: p.grad.float().clone() for n, p in m.named_parameters()})
    assert len(calls) == 2 and calls[0][2] == 8192                   # the first run took the kernel for both products
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
    assert rel(res[""][0], res["1"][0]) < 1e-4                        # (dB / dC / dA are atomic sums: runs differ in the last bits)
    touched = 0
    for n, g in res["1"][1].items():
        if "x_proj" in n or "dt_proj.weight" in n or "dt_proj_b.weight" in n or "dt_proj_s.weight" in n:
            assert rel(res[""][1][n], g) < 6e-3, n                    # bmm's 16-bit output rounding
            touched += 1
        else:
            assert rel(res[""][1][n], g) < 1e-4, n
    assert touched >= 2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,D,nf,hw", [(2, 8, 5, 48), (1, 3, 3, 8), (3, 64, 5, 256), (2, 4, 1, 64), (1, 2, 8, 8)])
def test_direction_maps_bit_exact(dtype, B, D, nf, hw, cuda):
    """csrc/dirmap.hip against the torch ops it replaces (mamba_simple.py:231, 245-247, 261-264): the stack is a pure
    permutation (bit-exact); the combine is (a + b + c) / 3 in fp32 (one rounding); both directions of autograd."""
    from vivim_amd.dirmap import combine_directions, stack_directions
    L = nf * hw
    g = torch.Generator().manual_seed(B * 100 + D)
    xz = torch.randn(2 * D, B, L, generator=g).to(dtype).to(cuda).transpose(0, 1)        # (L, B*L, 1)-strided like in_proj's
    xz.requires_grad_(True)
    stk = stack_directions(xz, nf)
    assert stk.shape == (B, 2, 3, D, L)
    xzv = xz.reshape(B, 2, D, L)
    ref_s = xzv.reshape(B, 2, D, nf, hw).transpose(3, 4).reshape(B, 2, D, L)
    assert torch.equal(stk[:, :, 0], xzv) and torch.equal(stk[:, :, 1], xzv.flip(-1)) and torch.equal(stk[:, :, 2], ref_s)
    w = torch.randn(B, 2, 3, D, L, generator=g).to(dtype).to(cuda)
    (gx,) = torch.autograd.grad(stk, xz, w)
    wf = w.float()
    ref_g = wf[:, :, 0] + wf[:, :, 1].flip(-1) + wf[:, :, 2].reshape(B, 2, D, hw, nf).transpose(3, 4).reshape(B, 2, D, L)
    assert torch.equal(gx.reshape(B, 2, D, L), ref_g.to(dtype))
    o3 = torch.randn(B, 3, D, L, generator=g).to(dtype).to(cuda).requires_grad_(True)
    y = combine_directions(o3, nf)
    of = o3.detach().float()
    ref_y = (of[:, 0] + of[:, 1].flip(-1) + of[:, 2].reshape(B, D, hw, nf).transpose(2, 3).reshape(B, D, L)) * (1.0 / 3.0)
    assert (y.float() - ref_y).abs().max() <= 2 * torch.finfo(dtype).eps * ref_y.abs().max()
    gy = torch.randn(B, D, L, generator=g).to(dtype).to(cuda)
    (go,) = torch.autograd.grad(y, o3, gy)
    r = (gy.float() * (1.0 / 3.0)).to(dtype)
    assert torch.equal(go[:, 0], r) and torch.equal(go[:, 1], r.flip(-1))
    assert torch.equal(go[:, 2], r.reshape(B, D, nf, hw).transpose(2, 3).reshape(B, D, L))


def test_grouped_module_random_shapes(cuda, monkeypatch):
    """Random v3 blocks (width, state size, frame count, spatial size, batch; fp32) through the grouped path and through
    the reference's three-call composition: same output, same input gradient, same parameter gradients.  Token counts
    that are not a multiple of 8 take the three-call composition in both runs (and must still agree trivially)."""
    hyp = pytest.importorskip("hypothesis")
    st = hyp.strategies
    from mamba_ssm import Mamba

    @hyp.settings(max_examples=25, deadline=None, derandomize=True, suppress_health_check=list(hyp.HealthCheck))
    @hyp.given(batch=st.integers(1, 3), d_model=st.sampled_from([8, 16, 32, 64]), d_state=st.sampled_from([4, 8, 16]),
               nf=st.sampled_from([1, 2, 3, 5]), hw=st.sampled_from([8, 16, 24, 64]), seed=st.integers(0, 999))
    def run(batch, d_model, d_state, nf, hw, seed):
        torch.manual_seed(seed)
        m = Mamba(d_model=d_model, d_state=d_state, d_conv=4, expand=2, bimamba_type="v3", nframes=nf).to(cuda)
        x = torch.randn(batch, nf * hw, d_model, device=cuda)
        res = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("VIVIM_SEPARATE_DIRECTIONS", mode)
            m.zero_grad(set_to_none=True)
            xi = x.clone().requires_grad_(True)
            y = m(xi)
            y.square().mean().backward()
            res[mode] = [y.detach(), xi.grad] + [p.grad.clone() for _, p in sorted(m.named_parameters())]
        for a, b in zip(res["0"], res["1"]):
            assert float((a - b).abs().max()) <= 3e-5 * float(b.abs().max()) + 1e-9

    run()


def test_fused_inner_op_reference_test_shape(cuda):
    """The shape of the reference's own test_mamba_inner_fn (tests/ops/test_selective_scan.py:152-249: dim 768,
    dstate 8, dt_rank 48, seqlen 128, batch 2) against the torch restatement of mamba_inner_ref, forward and every
    gradient (the reference asserts the forward only and prints the gradient differences)."""
    from mamba_ssm.ops.selective_scan_interface import mamba_inner_fn_no_out_proj
    from oracle import ref_torch
    b, d_inner, n, r, L = 2, 768, 8, 48, 128
    g = torch.Generator().manual_seed(0)
    p = dict(conv_w=torch.randn(d_inner, 1, 4, generator=g) * 0.3, conv_b=torch.randn(d_inner, generator=g) * 0.1,
             x_proj=torch.randn(r + 2 * n, d_inner, generator=g) * d_inner ** -0.5,
             dt_proj=(torch.rand(d_inner, r, generator=g) * 2 - 1) * r ** -0.5,
             A=-torch.rand(d_inner, n, generator=g) - 0.1, D=torch.randn(d_inner, generator=g),
             dt_bias=torch.rand(d_inner, generator=g) - 4.0)
    xz = torch.randn(b, 2 * d_inner, L, generator=g)
    dout = torch.randn(b, d_inner, L, generator=g)

    def run(device, fn, **kw):
        q = {k: v.to(device).requires_grad_(True) for k, v in p.items()}
        x = xz.to(device).requires_grad_(True)
        y = fn(x, q["conv_w"], q["conv_b"], q["x_proj"], q["dt_proj"], q["A"], *kw.get("mid", ()), q["D"],
               delta_bias=q["dt_bias"], delta_softplus=True)
        y.backward(dout.to(device))
        return [y.detach().cpu(), x.grad.cpu()] + [q[k].grad.cpu() for k in sorted(q)]

    got = run(cuda, mamba_inner_fn_no_out_proj, mid=(None, None))
    want = run("cpu", ref_torch.mamba_inner_no_out_proj_ref)
    for i, (a, w) in enumerate(zip(got, want)):
        assert rel_err(a, w) < 3e-4, i


def test_train_trajectory_grouped_equals_separate(cuda, monkeypatch):
    """Three full Vivim train steps (bf16 autocast, AdamW) from the same seed through the grouped three-direction path
    and through the reference's three-call composition: the RNG-consuming ops (DropPath, dropout, the CPU coin flips of
    the decode head) are the same torch ops in the same order in both, so the loss trajectories must agree to bf16
    round-off."""
    from vivim_amd.train_step import build_model, make_optimizer, synthetic_batch, train_step
    losses = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("VIVIM_SEPARATE_DIRECTIONS", mode)
        torch.manual_seed(11)
        model = build_model(3, cuda, mamba_kwargs={"d_state": 16, "expand": 2})
        opt = make_optimizer(model)
        clip, onehot = synthetic_batch(1, 5, 128, 3, cuda, 5)
        torch.manual_seed(12)
        losses[mode] = [float(train_step(model, opt, clip, onehot, 3, torch.bfloat16)) for _ in range(3)]
        del model, opt
    for a, b in zip(losses["0"], losses["1"]):
        assert abs(a - b) <= 5e-3 * abs(b), (losses["0"], losses["1"])


@pytest.mark.gpu
def test_lean_adamw_equals_torch_fused_adamw(cuda):
    """train_step.LeanFusedAdamW drives the same fused kernel as torch.optim.AdamW(fused=True) with cached lists:
    identical parameters after several steps, including a step in which one parameter has no gradient."""
    from vivim_amd.train_step import LeanFusedAdamW
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 32), (7,), (3, 5, 2), (1,), (128, 128)]
    pa = [torch.randn(*s, generator=g).to(cuda).requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    oa = LeanFusedAdamW(pa, lr=1e-2, betas=(0.9, 0.999), weight_decay=1e-2)
    ob = torch.optim.AdamW(pb, lr=1e-2, betas=(0.9, 0.999), weight_decay=1e-2, fused=True)
    for step in range(5):
        grads = [torch.randn(*s, generator=g).to(cuda) for s in shapes]
        for i, (a, b, gr) in enumerate(zip(pa, pb, grads)):
            skip = step == 2 and i == 1
            a.grad = None if skip else gr.clone()
            b.grad = None if skip else gr.clone()
        oa.step()
        ob.step()
        oa.zero_grad()
        ob.zero_grad(set_to_none=True)
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_train_step_fp16_scaling_clipping_and_optimizer_state(cuda):
    """The harness side of the reference's Trainer(precision=16, gradient_clip_val) (multiclass_training_folds.py:800-811):
    fp16 autocast steps under the dynamic loss scale stay finite and move the loss, clipping bounds the update, and the lean
    optimizer's state survives a state_dict round trip (same parameters after the same further step)."""
    import copy
    from vivim_amd.train_step import _SCALERS, build_model, make_optimizer, synthetic_batch, train_step
    torch.manual_seed(21)
    model = build_model(3, cuda, mamba_kwargs={"d_state": 16, "expand": 2}, drop_path_rate=0.0)
    opt = make_optimizer(model)
    clip, onehot = synthetic_batch(1, 3, 64, 3, cuda, 9)
    losses = [float(train_step(model, opt, clip, onehot, 3, torch.float16, clip_grad_norm=1.0)) for _ in range(3)]
    assert all(l == l and abs(l) < 1e4 for l in losses) and _SCALERS[id(opt)]["scale"] > 0
    assert all(torch.isfinite(p).all() for p in model.parameters())
    # state round trip: clone the model and the optimizer state, apply the SAME gradients to both (the kernels' fp32 atomics
    # make two backward passes differ in the last bits, and Adam's m / sqrt(v) amplifies that): identical parameters after
    model2 = copy.deepcopy(model)
    opt2 = make_optimizer(model2)
    opt2.load_state_dict(copy.deepcopy(opt.state_dict()))
    g = torch.Generator(device="cpu").manual_seed(44)
    for a, b in zip(model.parameters(), model2.parameters()):
        if a.requires_grad:
            a.grad = (1e-3 * torch.randn(a.shape, generator=g)).to(cuda)
            b.grad = a.grad.clone()
    opt.step()
    opt2.step()
    for (n, a), (_, b) in zip(model.named_parameters(), model2.named_parameters()):
        assert torch.equal(a, b), n
