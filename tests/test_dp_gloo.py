"""CPU, world_size 2, gloo: the data-parallel plumbing (vivim_amd/dp.py) that bench.py uses around the train
step -- rank-offset shards, DDP gradient averaging equal to the large-batch gradient, max-over-ranks timing,
and that freezing the never-used parameters leaves a graph DDP accepts without unused-parameter search."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class TinyVivimShape(torch.nn.Module):
    """Same *structure* as Vivim for DDP purposes: used trunk + two parameter groups the forward never touches."""

    def __init__(self):
        super().__init__()
        self.trunk = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.GELU(), torch.nn.Linear(16, 3))
        self.decoder = torch.nn.Module()
        self.decoder.classifier = torch.nn.Linear(16, 150)           # never used (vivim.py:325)
        self.encoder = torch.nn.Module()
        self.encoder.downsample_layers = torch.nn.Module()
        self.encoder.downsample_layers.layer_norm = torch.nn.ModuleList([torch.nn.LayerNorm(16)])   # never used

    def forward(self, x):
        return self.trunk(x)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from vivim_amd import dp
    w, r, _ = dp.init(backend="gloo")
    assert (w, r) == (world, rank)
    torch.manual_seed(7)                                   # identical replicas
    model = TinyVivimShape()
    assert dp.freeze_unused(model) == 4
    ddp = dp.wrap(model)
    g = torch.Generator().manual_seed(dp.shard_seed(100, rank))
    x, y = torch.randn(4, 8, generator=g), torch.randn(4, 3, generator=g)
    loss = torch.nn.functional.mse_loss(ddp(x), y)
    loss.backward()
    dp.barrier()

    def one_step():
        ddp.zero_grad()
        torch.nn.functional.mse_loss(ddp(x), y).backward()
    probe = dp.comm_probe(ddp, one_step, sync_ms_per_step=5.0)          # what bench.py reports under torchrun
    assert probe["n_ranks_seen"] == world and probe["allreduce_ms"] > 0 and 0.0 <= probe["overlap_frac"] <= 1.0
    assert probe["allreduce_MB"] >= 0 and probe["ms_per_step_no_exchange"] > 0
    ddp.zero_grad()
    torch.nn.functional.mse_loss(ddp(x), y).backward()                  # all-reduced again after the unsynchronised steps
    t = dp.max_over_ranks(1.0 + rank)
    grads = torch.cat([p.grad.flatten() for p in model.parameters() if p.requires_grad])
    torch.save({"grads": grads, "x": x, "y": y, "t": t}, os.path.join(out, f"r{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_ddp_gloo_world2(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"r{i}.pt") for i in range(2))
    assert not torch.equal(r0["x"], r1["x"])                               # different shards
    assert torch.allclose(r0["grads"], r1["grads"])                        # all-reduced gradients agree
    assert r0["t"] == r1["t"] == 2.0                                        # max over ranks
    torch.manual_seed(7)
    ref = TinyVivimShape()
    x, y = torch.cat([r0["x"], r1["x"]]), torch.cat([r0["y"], r1["y"]])
    torch.nn.functional.mse_loss(ref.trunk(x), y).backward()
    want = torch.cat([p.grad.flatten() for p in ref.trunk.parameters()])
    assert torch.allclose(r0["grads"], want, atol=1e-6)                    # = gradient of the global batch


def _mamba_worker(rank, world, port, out):
    """The REAL v3 Mamba module (three directions' parameters = views of fused (3, ...) storage) inside a 2-rank gloo DDP.
    No kernel runs (there is no GPU here and the product has no CPU path): the loss is a rank-dependent linear form of
    the parameters, which is all DDP's reducer, its buckets and the parameter broadcast ever see."""
    sys.path.insert(0, ROOT)
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from vivim_amd import dp
    from vivim_amd.mamba_simple import Mamba
    dp.init(backend="gloo")
    torch.manual_seed(11 + rank)                           # DIFFERENT initial replicas: DDP's broadcast must repair them
    m = Mamba(d_model=32, d_state=16, bimamba_type="v3", nframes=3)
    m._fuse()
    ptr = {n: p.data_ptr() for n, p in m.named_parameters()}
    ddp = dp.wrap(m)
    # (ii) the views survive DDP's construction (parameter broadcast from rank 0, bucket views)
    assert {n: p.data_ptr() for n, p in m.named_parameters()} == ptr
    for k, buf in enumerate(m._fused):
        if buf is not None:
            assert buf[0].data_ptr() == [m.conv1d.weight, m.conv1d.bias, m.x_proj.weight, m.dt_proj.weight, m.dt_proj.bias,
                                         m.A_log, m.D][k].data_ptr()
    w0 = torch.cat([p.detach().flatten() for p in m.parameters()])
    fused0 = torch.cat([b.flatten() for b in m._fused if b is not None])

    class Lin(torch.nn.Module):                            # forward through DDP: loss = sum_p <p, c_rank>
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, scale):
            return sum((p * (scale * (i % 5 + 1))).sum() for i, p in enumerate(self.inner.parameters()))
    ddp = dp.wrap(Lin(m))
    ddp(float(rank + 1)).backward()
    assert {n: p.data_ptr() for n, p in m.named_parameters()} == ptr      # gradient_as_bucket_view moved no parameter
    grads = {n: p.grad.clone() for n, p in m.named_parameters()}
    assert all(g is not None for g in grads.values())                     # (i) every Parameter view got a gradient
    torch.save({"w0": w0, "fused0": fused0, "grads": grads}, os.path.join(out, f"m{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_real_mamba_fused_views_in_ddp_world2(tmp_path):
    """multi-rank readiness of the fused parameter storage without a node."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_mamba_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"m{i}.pt") for i in range(2))
    assert torch.equal(r0["w0"], r1["w0"]) and torch.equal(r0["fused0"], r1["fused0"])    # broadcast reached the fused buffers
    for i, (n, g) in enumerate(r0["grads"].items()):
        assert torch.equal(g, r1["grads"][n])
        assert torch.allclose(g, torch.full_like(g, 1.5 * (i % 5 + 1))), n             # (iii) the two-rank mean of (1, 2) * c


def test_single_process_helpers_are_identity():
    sys.path.insert(0, ROOT)
    from vivim_amd import dp
    m = torch.nn.Linear(2, 2)
    assert dp.wrap(m) is m and dp.max_over_ranks(3.5) == 3.5 and dp.shard_seed(1, 3) == 4


def test_freeze_unused_on_the_real_parameter_list():
    """The real Vivim (random-init SegFormer-b3 backbone, CPU, no forward): freeze_unused must freeze exactly the two
    parameter groups the forward never reaches (modeling/vivim.py:211-212, 325) -- what DDP then reduces is every other
    parameter.  (tests/test_gpu_model.py checks on the GPU that each of those does receive a gradient.)"""
    sys.path.insert(0, ROOT)
    from vivim_amd.train_step import build_model
    model = build_model(3, "cpu")
    frozen = sorted(n for n, p in model.named_parameters() if not p.requires_grad)
    assert frozen, "nothing frozen"
    assert all(n.startswith("decoder.classifier.") or ".layer_norm." in n and n.startswith("encoder.downsample_layers.")
               for n in frozen), frozen
    live = [n for n, p in model.named_parameters() if p.requires_grad]
    assert not any(n.startswith("decoder.classifier.") for n in live)
    assert len(live) > 500                                               # SegFormer-b3 + 8 Mamba layers + head
