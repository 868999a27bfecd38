"""Data-parallel plumbing of the train step: one process per GPU, RCCL (backend "nccl")
on MI355X / gloo on CPU, identical replicas, rank-offset synthetic shards, bucketed gradient all-reduce by DDP,
max-over-ranks timing.  Nothing here touches the kernels: clips are independent, the only exchange is the
gradient all-reduce."""
import os

import torch
import torch.distributed as dist


# VIVIM_DP_REHEARSE=1: run the whole multi-rank code path (process group on RCCL, DDP buckets and hooks, all-reduced
# timing, barriers) with a world of ONE rank -- the only way to exercise it on a one-GPU box.
_REHEARSE = os.environ.get("VIVIM_DP_REHEARSE", "0") == "1"


def dist_env():
    """-> (world, rank, local_rank) from the torchrun environment (1, 0, 0 when launched plainly)."""
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend=None, device=None):
    """Initialise the default process group when WORLD_SIZE > 1.  backend defaults to nccl (= RCCL over xGMI
    on ROCm) when a GPU device is given, else gloo."""
    world, rank, local_rank = dist_env()
    if (world > 1 or _REHEARSE) and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if _REHEARSE:
            for k, v in (("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
                os.environ.setdefault(k, v)
        backend = backend or ("nccl" if device is not None and device.type == "cuda" else "gloo")
        kw = {"device_id": device} if backend == "nccl" else {}
        dist.init_process_group(backend, **kw)
    return world, rank, local_rank


def freeze_unused(model):
    """Freeze parameters that never receive a gradient in Vivim's forward (the SegFormer 150-class classifier and
    the per-stage encoder norms, modeling/vivim.py:211-212, 325) so DDP needs no unused-parameter search."""
    n = 0
    dec = getattr(model, "decoder", None)
    if dec is not None and hasattr(dec, "classifier"):
        for p in dec.classifier.parameters():
            p.requires_grad_(False)
            n += 1
    enc = getattr(getattr(model, "encoder", None), "downsample_layers", None)
    if enc is not None and hasattr(enc, "layer_norm"):
        for p in enc.layer_norm.parameters():
            p.requires_grad_(False)
            n += 1
    return n


def wrap(model, device=None, bucket_cap_mb=25):
    """DDP with 25 MB buckets in reverse registration order (the stage-3/2 buckets fly while the long stage-0/1
    backward scans still run); identity when the world is 1."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not _REHEARSE):
        return model
    ids = [device.index] if device is not None and device.type == "cuda" else None
    # a channels-last parameter (the decode head's 1x1 `out` conv) gets a channels-last gradient, which cannot be a view of
    # its contiguous bucket ("Grad strides do not match bucket view strides": an extra copy per step): store it densely
    for prm in model.parameters():
        if not prm.data.is_contiguous():
            prm.data = prm.data.contiguous()
        if prm.requires_grad and prm.dim() == 4 and prm.shape[2] == prm.shape[3] == 1:
            # MIOpen returns the weight gradient of a 1x1 convolution with channels-last strides on its two size-1 axes
            # ((768, 1, 768, 768) for a (3, 768, 1, 1) weight): the same bytes as the dense layout, but DDP compares the
            # stride tuples and copies.  Hand it the dense strides over the same storage.
            def dense_strides(g):
                want = tuple(torch.empty(g.shape, device="meta").stride())
                return g.as_strided(g.shape, want, g.storage_offset()) if g.is_contiguous() and g.stride() != want else g
            prm.register_hook(dense_strides)
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, bucket_cap_mb=bucket_cap_mb,
                                                     gradient_as_bucket_view=True)


def shard_seed(seed, rank):
    """Every rank draws its own synthetic clips (a DistributedSampler-equivalent for synthetic data)."""
    return seed + rank


def max_over_ranks(seconds, device=None):
    """The step time of the job is that of its slowest rank."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not _REHEARSE):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(device=None):
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _REHEARSE):
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def comm_probe(step_model, run_step, sync_ms_per_step, device=None, steps=3):
    """What the gradient exchange costs and how much of it the backward hides, measured after the
    timed region on a world of > 1 ranks:
      allreduce_ms  -- all the gradient bytes reduced again, alone, in bucket-sized pieces (max over ranks);
      overlap_frac  -- 1 - (step with exchange - step without) / allreduce_ms, clamped to [0, 1]; the step without exchange
                       runs under DDP.no_sync() (replicas drift apart afterwards: call this last).
    -> dict (empty when the world is one rank and this is no rehearsal)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not _REHEARSE):
        return {}
    import time
    out = {"n_ranks_seen": dist.get_world_size()}
    grads = [p for p in step_model.parameters() if p.requires_grad]
    nbytes = sum(p.numel() * p.element_size() for p in grads)
    cap = 25 * 1024 * 1024 // 4
    flat = torch.zeros(sum(p.numel() for p in grads), dtype=torch.float32, device=grads[0].device)
    pieces = list(flat.split(cap))
    cuda = device is not None and device.type == "cuda"

    def once():
        barrier(device)
        t0 = time.perf_counter()
        for t in pieces:
            dist.all_reduce(t)
        if cuda:
            torch.cuda.synchronize(device)
        return time.perf_counter() - t0
    once()
    ar = min(once() for _ in range(3))
    out["allreduce_ms"] = round(max_over_ranks(ar, device) * 1e3, 3)
    out["allreduce_MB"] = round(nbytes / 1e6, 1)
    if hasattr(step_model, "no_sync"):
        barrier(device)
        t0 = time.perf_counter()
        with step_model.no_sync():
            for _ in range(steps):
                run_step()
        barrier(device)
        nosync_ms = max_over_ranks((time.perf_counter() - t0) / steps, device) * 1e3
        out["ms_per_step_no_exchange"] = round(nosync_ms, 3)
        exposed = max(0.0, sync_ms_per_step - nosync_ms)
        out["overlap_frac"] = round(min(1.0, max(0.0, 1.0 - exposed / max(out["allreduce_ms"], 1e-6))), 4)
    return out
