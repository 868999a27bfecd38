#!/usr/bin/env python3
"""Which part of the train step is not replay-stable under HIP-graph capture?  Captures components separately,
replays each three times on unchanged inputs and reports whether the outputs repeat."""
import os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vivim_amd.train_step import build_model, synthetic_batch
from vivim_amd.vivim import MambaLayer
from mamba_ssm import Mamba
dev = torch.device("cuda:0")
torch.manual_seed(0)

def probe(name, fn, warm=2):
    try:
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warm): fn()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            outs = fn()
        res = []
        for _ in range(3):
            g.replay(); torch.cuda.synchronize()
            res.append([o.detach().float().clone() for o in outs])
        diffs = [max(float((a - b).abs().max()) for a, b in zip(res[0], r)) for r in res[1:]]
        fin = all(bool(torch.isfinite(o).all()) for r in res for o in r)
        print(f"[probe] {name}: replay diffs {diffs} finite={fin}", flush=True)
    except Exception as e:
        print(f"[probe] {name}: FAIL {type(e).__name__}: {str(e).splitlines()[0]}", flush=True)
        torch.cuda.synchronize()

amp = lambda: torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False)
# 1. Mamba module fwd / fwd+bwd
m = Mamba(d_model=64, bimamba_type="v3").to(dev)
xm = torch.randn(3, 5 * 64 * 64, 64, device=dev, requires_grad=True)
def mamba_fwd():
    with amp(): return [m(xm, nframes=5)]
probe("Mamba fwd (L=20480)", mamba_fwd)
def mamba_fb():
    for p in m.parameters(): p.grad = None
    xm.grad = None
    with amp(): y = m(xm, nframes=5)
    y.float().square().mean().backward()
    return [y, xm.grad] + [p.grad for p in m.parameters()]
probe("Mamba fwd+bwd (L=20480)", mamba_fb)
# 2. MambaLayer (adds LN, Mlp with dwconv)
layer = MambaLayer(64).to(dev).eval()
xl = torch.randn(3, 64, 5, 64, 64, device=dev, requires_grad=True)
def layer_fb():
    for p in layer.parameters(): p.grad = None
    xl.grad = None
    with amp(): y = layer(xl)
    y.float().square().mean().backward()
    return [y, xl.grad] + [p.grad for p in layer.parameters()]
probe("MambaLayer fwd+bwd", layer_fb)
# 3. SegFormer encoder stage 0 only (HF code) and full model forward in eval
model = build_model(3, dev).eval()
clip, onehot = synthetic_batch(3, 5, 256, 3, dev, 0)
enc = model.encoder.downsample_layers
def seg_stage0():
    with amp():
        hs, h, w = enc.patch_embeddings[0](clip.reshape(15, 3, 256, 256))
        for blk in enc.block[0]:
            hs = blk(hs, h, w)
            hs = hs[0] if isinstance(hs, (tuple, list)) else hs
    return [hs]
probe("SegFormer stage 0 fwd", seg_stage0)
def full_fwd():
    with amp(): return [model(clip)]
probe("Vivim fwd (eval)", full_fwd)
def full_fb():
    for p in model.parameters(): p.grad = None
    with amp(): y = model(clip)
    y.float().square().mean().backward()
    return [y] + [p.grad for p in model.parameters() if p.grad is not None][:40]
probe("Vivim fwd+bwd (eval)", full_fb)
# 4. train mode, with and without stochastic layers
from vivim_amd.train_step import recall_focused_loss
alpha = torch.tensor([0.05, 0.475, 0.475], device=dev)
oh = onehot.reshape(15, 3, 256, 256)
for tag, kill in (("train, stochastic layers on", False), ("train, all dropout p=0", True)):
    torch.manual_seed(0)
    mdl = build_model(3, dev, drop_path_rate=0.0 if kill else 0.2).train()
    if kill:
        mdl.dropout_rate = 0.0
        for mod in mdl.modules():
            if isinstance(mod, (torch.nn.Dropout, torch.nn.Dropout2d)): mod.p = 0.0
    def tr_fb():
        for p in mdl.parameters(): p.grad = None
        with amp(): y = mdl(clip)
        loss = recall_focused_loss(y, None, 3, onehot=oh, alpha=alpha)
        loss.backward()
        return [loss.detach().reshape(1), y] + [p.grad for p in mdl.parameters() if p.grad is not None]
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): tr_fb()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = tr_fb()
    names = ["loss", "logits"] + [n for n, p in mdl.named_parameters() if p.requires_grad]
    for it in range(3):
        g.replay(); torch.cuda.synchronize()
        bad = [names[i] for i, o in enumerate(outs) if not torch.isfinite(o).all()]
        print(f"[probe] {tag}: replay {it} loss {float(outs[0]):.6f} non-finite: {len(bad)} {bad[:6]}", flush=True)
