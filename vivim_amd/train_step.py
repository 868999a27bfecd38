"""Plain-PyTorch counterpart of the reference's Lightning train step (multiclass_training_folds.py:543-573,
configure_optimizers :503-517, Trainer(precision=16, devices=1) :800-811): Vivim forward under autocast,
recall_focused_loss, backward, AdamW.  Used by bench.py and the tests; the losses are restated from the
reference source text (it cannot be imported: it needs pytorch_lightning / wandb / medpy and parses argv at
import time)."""
import torch
import torch.nn.functional as F


def tversky_loss(probs, onehot, alpha=0.3, beta=0.7, smooth=1e-6):
    """multiclass_training_folds.py:218-255: per class and per image, then mean over images and classes."""
    tp = (probs * onehot).sum(dim=(2, 3))
    fp = (probs * (1 - onehot)).sum(dim=(2, 3))
    fn = ((1 - probs) * onehot).sum(dim=(2, 3))
    tv = (tp + smooth) / (tp + alpha * fp + beta * fn + smooth)          # (N, C)
    return (1 - tv.mean(dim=0)).mean()


_ALPHA = {}       # (values, device, dtype) -> device tensor: built once, a host->device copy per step would stall the host


def _alpha_tensor(alpha, device, dtype):
    key = (tuple(alpha), str(device), dtype)
    t = _ALPHA.get(key)
    if t is None:
        t = _ALPHA[key] = torch.tensor(alpha, device=device, dtype=dtype)
    return t


def class_balanced_focal_loss(probs, onehot, gamma=2.0, alpha=(0.05, 0.475, 0.475)):
    """multiclass_training_folds.py:363-423 with explicit alpha."""
    a = alpha if torch.is_tensor(alpha) else _alpha_tensor(alpha, probs.device, probs.dtype)
    a = a[None, :, None, None]
    weight = onehot * (1 - probs) ** gamma + (1 - onehot) * probs ** gamma
    bce = -onehot * torch.log(probs + 1e-6) - (1 - onehot) * torch.log(1 - probs + 1e-6)
    return (a * weight * bce).mean(dim=(0, 2, 3)).sum()


def recall_focused_loss(logits, targets, num_classes, gamma=2.0, onehot=None, alpha=None):
    """0.4 * focal + 0.6 * tversky (multiclass_training_folds.py:339-361). logits (N,C,H,W); targets (N,H,W).
    `onehot` (N,C,H,W float) and `alpha` (device tensor) may be passed pre-built (graph capture: no host->device
    copy may happen inside the captured region)."""
    probs = F.softmax(logits.float(), dim=1)
    if onehot is None:
        onehot = F.one_hot(targets.long(), num_classes).permute(0, 3, 1, 2).float()
    if alpha is None:
        alpha = (0.05, 0.475, 0.475) if num_classes == 3 else tuple([1.0 / num_classes] * num_classes)
    return 0.4 * class_balanced_focal_loss(probs, onehot, gamma, alpha) + 0.6 * tversky_loss(probs, onehot)


def build_model(num_classes=3, device="cuda", mamba_kwargs=None, drop_path_rate=0.2, fast_backbone_dwconv=True):
    """Vivim with a randomly initialised SegFormer-b3 backbone (no hub access on the GPU box).  The two
    parameter groups that never receive a gradient in Vivim's forward -- the SegFormer 150-class classifier
    and the per-stage encoder layer norms (vivim.py:211-212, 325) -- are frozen so DDP needs no
    unused-parameter search."""
    from .vivim import Vivim, segformer_b3_random
    model = Vivim(in_chans=3, out_chans=num_classes, backbone=segformer_b3_random(),
                  drop_path_rate=drop_path_rate, mamba_kwargs=mamba_kwargs,
                  fast_backbone_dwconv=fast_backbone_dwconv)
    from .dp import freeze_unused
    freeze_unused(model)
    return model.to(device)


class LeanFusedAdamW:
    """AdamW on PyTorch's fused multi-tensor kernel (`torch._fused_adamw_`, the kernel `torch.optim.AdamW(fused=True)`
    launches) with the per-step Python of `torch.optim` taken out: the parameter / state lists are built once, a step is
    one `_foreach_add_` on the step counters and one fused launch per (device, dtype) group.  `torch.optim.AdamW.step`
    re-derives those lists from the param groups and their state dicts on every call (~2.6 ms of host time for Vivim's
    600 tensors, in a step whose critical path is the host).  Same arithmetic, same hyper-parameters."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.params = [p for p in params if p.requires_grad]
        assert self.params and all(p.is_cuda and p.dtype == torch.float32 for p in self.params)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.groups = {}
        for p in self.params:
            self.groups.setdefault(p.device, []).append(p)
        self.state = {dev: ([torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps],
                            [torch.zeros((), dtype=torch.float32, device=dev) for _ in ps])
                      for dev, ps in self.groups.items()}

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    def state_dict(self):
        """Checkpointable state: hyper-parameters and (exp_avg, exp_avg_sq, step) per parameter, in parameter order."""
        out = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay, "state": []}
        where = {id(p): (dev, i) for dev, ps in self.groups.items() for i, p in enumerate(ps)}
        for p in self.params:
            dev, i = where[id(p)]
            m, v, t = self.state[dev]
            out["state"].append({"exp_avg": m[i], "exp_avg_sq": v[i], "step": t[i]})
        return out

    def load_state_dict(self, sd):
        assert len(sd["state"]) == len(self.params), "optimizer state does not match the parameter list"
        self.lr, self.betas, self.eps, self.weight_decay = sd["lr"], tuple(sd["betas"]), sd["eps"], sd["weight_decay"]
        where = {id(p): (dev, i) for dev, ps in self.groups.items() for i, p in enumerate(ps)}
        with torch.no_grad():
            for p, st in zip(self.params, sd["state"]):
                dev, i = where[id(p)]
                m, v, t = self.state[dev]
                m[i].copy_(st["exp_avg"]); v[i].copy_(st["exp_avg_sq"]); t[i].copy_(st["step"])

    @torch.no_grad()
    def step(self):
        for dev, ps in self.groups.items():
            exp_avg, exp_avg_sq, steps = self.state[dev]
            grads = [p.grad for p in ps]
            if any(g is None for g in grads):                                 # a parameter without a gradient this step
                keep = [i for i, g in enumerate(grads) if g is not None]
                ps, grads = [ps[i] for i in keep], [grads[i] for i in keep]
                exp_avg, exp_avg_sq, steps = ([exp_avg[i] for i in keep], [exp_avg_sq[i] for i in keep],
                                              [steps[i] for i in keep])
            torch._foreach_add_(steps, 1)
            torch._fused_adamw_(ps, grads, exp_avg, exp_avg_sq, [], steps, lr=self.lr, beta1=self.betas[0],
                                beta2=self.betas[1], weight_decay=self.weight_decay, eps=self.eps, amsgrad=False,
                                maximize=False)


def make_optimizer(model, lr=1e-4, weight_decay=1e-2, capturable=False, fused=None, lean=None):
    """AdamW(lr 1e-4, betas (.9, .999), wd 1e-2) as the reference's configure_optimizers (train.py:503-517).  On a GPU
    the update runs as PyTorch's fused multi-tensor kernel (same arithmetic, one launch per dtype/device group
    instead of a Python-side foreach chain: the optimizer was ~10 ms of the host-bound 80 ms step), by default through
    `LeanFusedAdamW`; `lean=False` gives `torch.optim.AdamW(fused=True)`."""
    params = [p for p in model.parameters() if p.requires_grad]
    on_gpu = bool(params) and all(p.is_cuda and p.dtype == torch.float32 for p in params)
    if fused is None:
        fused = on_gpu and not capturable
    if lean is None:
        lean = fused and on_gpu
    if lean:
        return LeanFusedAdamW(params, lr=lr, betas=(0.9, 0.999), weight_decay=weight_decay)
    kw = {"fused": True} if fused else {"capturable": capturable}
    return torch.optim.AdamW(params, lr=lr, betas=(0.9, 0.999), weight_decay=weight_decay, **kw)


def synthetic_batch(batch, clip_length, image_size, num_classes, device, seed):
    """clip ~ N(0,1) (the reference normalises frames with ImageNet mean/std, Multiclass_Data.py:24-25);
    one-hot float target (B, nf, C, H, W) as the dataset yields (Multiclass_Data.py:211-215)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    clip = torch.randn(batch, clip_length, 3, image_size, image_size, generator=g)
    labels = torch.randint(0, num_classes, (batch, clip_length, image_size, image_size), generator=g)
    onehot = F.one_hot(labels, num_classes).permute(0, 1, 4, 2, 3).float()
    return clip.to(device), onehot.to(device)


_SCALERS = {}


def train_step(model, optimizer, clip, onehot, num_classes, amp_dtype=torch.bfloat16, clip_grad_norm=None):
    """One fwd + loss + bwd + optimizer step; returns the detached loss.  fp16 autocast runs under a GradScaler (the
    reference trains with Trainer(precision=16), multiclass_training_folds.py:800-811; bf16 / fp32 need none);
    `clip_grad_norm` (the reference's gradient_clip_val) clips the global gradient norm before the update."""
    if not model.training:                 # Module.train() walks all ~4000 submodules: 3 ms of host time when repeated per step
        model.train()
    with torch.autocast("cuda", dtype=amp_dtype, enabled=amp_dtype != torch.float32):
        logits = model(clip)                                   # (B*nf, C, H, W)
    B, T = onehot.shape[:2]
    target = onehot.argmax(dim=2).view(B * T, *onehot.shape[-2:])
    loss = recall_focused_loss(logits, target, num_classes)
    optimizer.zero_grad(set_to_none=True)
    if amp_dtype == torch.float16:
        sc = _SCALERS.setdefault(id(optimizer), {"scale": 65536.0, "good": 0})      # dynamic loss scale, GradScaler's policy
        (loss * sc["scale"]).backward()
        params = [p for p in model.parameters() if p.grad is not None]
        torch._foreach_mul_([p.grad for p in params], 1.0 / sc["scale"])
        finite = bool(torch.stack([torch.isfinite(p.grad).all() for p in params]).all())   # one host sync per step
        if finite:
            if clip_grad_norm is not None:
                torch.nn.utils.clip_grad_norm_(params, clip_grad_norm)
            optimizer.step()
            sc["good"] += 1
            if sc["good"] == 2000:
                sc["scale"], sc["good"] = sc["scale"] * 2.0, 0
        else:                                                # skip the update, halve the scale
            sc["scale"], sc["good"] = sc["scale"] * 0.5, 0
        return loss.detach()
    loss.backward()
    if clip_grad_norm is not None:
        torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], clip_grad_norm)
    optimizer.step()
    return loss.detach()
