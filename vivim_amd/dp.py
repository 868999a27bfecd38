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
