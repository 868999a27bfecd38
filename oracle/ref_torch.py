"""Pure-PyTorch restatement of the reference's CPU-runnable oracles.

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline leg).  Autograd-capable, so it
also yields reference gradients for the fused op and the v3 module.  Pinned against the reference's
own functions through tests/golden (see tests/golden/make_golden.py and tests/test_oracle.py).

Restated from (file:line in /root/reference):
  selective_scan_ref          mamba/mamba_ssm/ops/selective_scan_interface.py:86-152
  causal_conv1d_ref           causal-conv1d/causal_conv1d/causal_conv1d_interface.py:49-65
  mamba_inner_ref (no o-proj) mamba/mamba_ssm/ops/selective_scan_interface.py:636-670 / 155-225
  Mamba.forward, v3 branch    mamba/mamba_ssm/modules/mamba_simple.py:188-264
"""
import torch
import torch.nn.functional as F


def selective_scan_ref(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                       return_last_state=False):
    """Sequential SSM recurrence in fp32 (real A only).

    u, delta, z: (b, d, l); A: (d, n); B, C: (d, n) | (b, n, l) | (b, g, n, l); D, delta_bias: (d,).
    """
    in_dtype = u.dtype
    u32 = u.float()
    dt = delta.float()
    if delta_bias is not None:
        dt = dt + delta_bias.float()[:, None]
    if delta_softplus:
        dt = F.softplus(dt)
    b, d, l = u32.shape
    n = A.shape[1]

    def per_channel(M):
        # -> (b, d, n, l) view/expansion of a variable B or C
        M = M.float()
        if M.dim() == 3:
            return M[:, None].expand(b, d, n, l)
        return M.repeat_interleave(d // M.shape[1], dim=1)

    var_b, var_c = B.dim() >= 3, C.dim() >= 3
    decay = torch.exp(dt[:, :, None, :] * A.float()[None, :, :, None])            # (b, d, n, l)
    if var_b:
        drive = per_channel(B) * (dt * u32)[:, :, None, :]
    else:
        drive = B.float()[None, :, :, None] * (dt * u32)[:, :, None, :]
    Cfull = per_channel(C) if var_c else None
    h = u32.new_zeros(b, d, n)
    ys = []
    for t in range(l):
        h = decay[..., t] * h + drive[..., t]
        if var_c:
            ys.append((h * Cfull[..., t]).sum(-1))
        else:
            ys.append((h * C.float()[None]).sum(-1))
    y = torch.stack(ys, dim=-1)
    if D is not None:
        y = y + u32 * D.float()[:, None]
    if z is not None:
        y = y * F.silu(z.float())
    y = y.to(in_dtype)
    return (y, h) if return_last_state else y


def causal_conv1d_ref(x, weight, bias=None, activation=None):
    """Depthwise causal conv, zero left pad, optional SiLU. x: (b, d, l); weight: (d, w)."""
    if activation not in (None, "silu", "swish"):
        raise NotImplementedError("activation must be None, silu, or swish")
    in_dtype = x.dtype
    d, w = weight.shape
    l = x.shape[-1]
    y = F.conv1d(x.to(weight.dtype), weight[:, None, :], bias, padding=w - 1, groups=d)[..., :l]
    if activation is not None:
        y = F.silu(y)
    return y.to(in_dtype)


def mamba_inner_no_out_proj_ref(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                                A, D=None, delta_bias=None, delta_softplus=True):
    """conv1d+SiLU -> x_proj -> dt_proj -> scan(+D, gate by silu(z)); returns (b, d_inner, l).
    Input-dependent B and C only (the Vivim call pattern, mamba_simple.py:217-228)."""
    l = xz.shape[-1]
    r = delta_proj_weight.shape[1]
    n = A.shape[-1]
    x, z = xz.chunk(2, dim=1)
    x = causal_conv1d_ref(x, conv1d_weight.squeeze(1), conv1d_bias, "silu")
    bsz, d_in, _ = x.shape
    x_dbl = F.linear(x.transpose(1, 2).reshape(bsz * l, d_in), x_proj_weight)     # (b l, r + 2n)
    delta = (delta_proj_weight @ x_dbl[:, :r].t()).reshape(d_in, bsz, l).transpose(0, 1)
    Bm = x_dbl[:, r:r + n].reshape(bsz, l, n).transpose(1, 2).contiguous()
    Cm = x_dbl[:, -n:].reshape(bsz, l, n).transpose(1, 2).contiguous()
    return selective_scan_ref(x, delta, A, Bm, Cm, D, z=z, delta_bias=delta_bias,
                              delta_softplus=delta_softplus)


def bimamba_inner_ref(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias,
                      A, A_b, D=None, delta_bias=None, delta_softplus=True):
    """selective_scan_interface.py:673-709 with input-dependent B and C: one conv / x_proj / dt_proj, the scan of the sequence
    with A plus the scan of the flipped sequence with A_b (flipped back), then out_proj; returns (b, l, d_model)."""
    l = xz.shape[-1]
    r = delta_proj_weight.shape[1]
    n = A.shape[-1]
    x, z = xz.chunk(2, dim=1)
    x = causal_conv1d_ref(x, conv1d_weight.squeeze(1), conv1d_bias, "silu")
    bsz, d_in, _ = x.shape
    x_dbl = F.linear(x.transpose(1, 2).reshape(bsz * l, d_in), x_proj_weight)
    delta = (delta_proj_weight @ x_dbl[:, :r].t()).reshape(d_in, bsz, l).transpose(0, 1)
    Bm = x_dbl[:, r:r + n].reshape(bsz, l, n).transpose(1, 2).contiguous()
    Cm = x_dbl[:, -n:].reshape(bsz, l, n).transpose(1, 2).contiguous()
    y = selective_scan_ref(x, delta, A, Bm, Cm, D, z=z, delta_bias=delta_bias, delta_softplus=delta_softplus)
    y_b = selective_scan_ref(x.flip([-1]), delta.flip([-1]), A_b, Bm.flip([-1]), Cm.flip([-1]), D, z=z.flip([-1]),
                             delta_bias=delta_bias, delta_softplus=delta_softplus)
    return F.linear((y + y_b.flip([-1])).transpose(1, 2), out_proj_weight, out_proj_bias)


def mamba_v3_forward_ref(hidden, p, nframes):
    """v3 tri-directional Mamba forward (mamba_simple.py:188-264) on a dict of parameters `p` with the
    module's state-dict names (in_proj.weight, conv1d{,_b,_s}.weight/bias, x_proj*.weight,
    dt_proj*.weight/bias, A{,_b,_s}_log, D{,_b,_s}, out_proj.weight)."""
    bsz, l, _ = hidden.shape
    xz = (p["in_proj.weight"] @ hidden.reshape(bsz * l, -1).t())
    d2 = xz.shape[0]
    xz = xz.reshape(d2, bsz, l).transpose(0, 1)                                   # (b, 2*d_inner, l)

    def inner(inp, sfx):
        return mamba_inner_no_out_proj_ref(
            inp, p[f"conv1d{sfx}.weight"], p[f"conv1d{sfx}.bias"], p[f"x_proj{sfx}.weight"],
            p[f"dt_proj{sfx}.weight"], -torch.exp(p[f"A{sfx}_log"].float()), p[f"D{sfx}"].float(),
            delta_bias=p[f"dt_proj{sfx}.bias"].float(), delta_softplus=True)

    out = inner(xz, "")
    out_b = inner(xz.flip([-1]), "_b").flip([-1])
    hw = l // nframes
    # frame-major (t*hw + p) -> pixel-major (p*nframes + t): mamba_simple.py:245-247, inverse :261
    xz_s = xz.reshape(bsz, d2, nframes, hw).transpose(2, 3).reshape(bsz, d2, l)
    out_s = inner(xz_s, "_s")
    out_s = out_s.reshape(bsz, -1, hw, nframes).transpose(2, 3).reshape(bsz, -1, l)
    y = (out + out_b + out_s).transpose(1, 2) / 3
    return F.linear(y, p["out_proj.weight"], p.get("out_proj.bias"))
