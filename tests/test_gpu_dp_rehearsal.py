"""bench.py's N > 1 path on real device tensors without an 8-GPU node: two ranks on the one GPU of the box, gradients reduced
by gloo through the host (VIVIM_DP_BACKEND=gloo) -- DDP's buckets over the fused (3, ...) parameter views, the custom autograd
nodes of the hot path under DDP hooks, rank-offset shards, max-over-ranks timing, the one JSON line.  (RCCL itself needs one GPU
per rank; the driver's scaling runs use it.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_gpu():
    env = dict(os.environ, VIVIM_DP_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    for port in ("29517", "29641"):                               # a second port if the first one is taken on the box
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
               "--warmup", "1", "--no-cpu-baseline", "--no-by-config"]
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        if r.returncode == 0 or "in use" not in r.stderr.lower():
            break
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 6 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["scaling"] == "weak" and d["steps"] == 2
    assert d["loss"] == d["loss"]                                # finite
    assert d["comm"]["n_ranks_seen"] == 2 and d["comm"]["allreduce_ms"] > 0
