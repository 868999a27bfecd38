#!/usr/bin/env python3
"""bench.py -- headline benchmark: frames/sec of one Vivim train step (fwd + loss + bwd + AdamW) at
256x256, clip_length 5, 3 classes, per-GPU batch 3, bf16 autocast (BASELINE.json configs[1]); weak scaling
over N GPUs (one process per GPU, RCCL gradient all-reduce overlapped with the backward by DDP buckets).

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      -- the dominant hot-path kernel (largest total time among the four C-ABI kernels), its
                   algorithmic bytes (vivim_amd/_lib.py:algorithmic_bytes) divided by its
                   HIP-event time, both summed over every launch inside the timed region;
                   `traffic` is the HBM bytes per launch from committed rocprofv3 --pmc passes over this command
                   (profiles/r*_bench_pmc_traffic.json), null when none is committed;
  cpu_baseline  -- the pure-PyTorch selective_scan_ref port (oracle/ref_torch.py) timed on this host's cores
                   on a bounded sample (N = 1 only).
Synthetic data, random-init weights (SegFormer-b3 architecture from a local config): there is no network.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s); the copy rate measured here is reported beside it


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)      # ~2.5 s timed: the step is host-bound and short runs are noisy
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("-image_size", "--image-size", type=int, default=256)
    ap.add_argument("-clip_length", "--clip-length", type=int, default=5)
    ap.add_argument("-train_bs", "--train-bs", type=int, default=3)
    ap.add_argument("-num_classes", "--num-classes", type=int, default=3)
    ap.add_argument("-seed", "--seed", type=int, default=42)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--d-state", type=int, default=16)
    ap.add_argument("--expand", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-by-config", action="store_true", help="skip the per-config kernel timings after the timed region")
    ap.add_argument("--stock-backbone-dwconv", action="store_true",
                    help="leave the SegFormer blocks stock: Mix-FFN depthwise convs on MIOpen instead of csrc/dwconv.hip, "
                         "transformers' own DropPath")
    return ap.parse_args()


def cpu_baseline(seconds_budget=12.0):
    """Times the pure-PyTorch selective_scan_ref port on the host CPUs: stage-0 scan shape of the benchmarked
    config at batch 1 (D=128, N=16, L=20480), fp32, forward only -- the reference's CPU-runnable path
    (BASELINE.json configs[0]).  Repeated for about `seconds_budget` seconds of CPU work (at
    least 3 calls); reported in algorithmic GB/s like the roofline, from the best call."""
    from oracle import ref_torch
    D, N, L = 128, 16, 20480
    g = torch.Generator().manual_seed(0)
    u = torch.randn(1, D, L, generator=g)
    delta = 0.5 * torch.rand(1, D, L, generator=g)
    A = -0.5 * torch.rand(D, N, generator=g)
    Bm, Cm = torch.randn(1, N, L, generator=g), torch.randn(1, N, L, generator=g)
    Dv, z = torch.randn(D, generator=g), torch.randn(1, D, L, generator=g)
    bias = 0.5 * torch.rand(D, generator=g)
    cores = torch.get_num_threads()
    times = []
    t_start = time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - t_start) < seconds_budget:
        t0 = time.perf_counter()
        with torch.no_grad():
            ref_torch.selective_scan_ref(u, delta, A, Bm, Cm, Dv, z=z, delta_bias=bias, delta_softplus=True)
        times.append(time.perf_counter() - t0)
        if len(times) >= 64:
            break
    best, mean = min(times), sum(times) / len(times)
    nbytes = 5 * D * L * 4 + 2 * N * L * 4 + 4 * (D * N + 2 * D)
    return {"value": round(nbytes / best / 1e9, 5), "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": f"selective_scan_ref port fwd, (B,D,N,L)=(1,{D},{N},{L}) fp32, {len(times)} calls in "
                      f"{sum(times):.1f} s, best {best:.2f} s (mean {mean:.2f} s) = {L / best:.0f} tokens/s on "
                      f"{os.cpu_count()} host cpus",
            "config1_mamba_forward": cpu_config1_mamba(ref_torch)}


def cpu_config1_mamba(ref_torch):
    """BASELINE.json configs[0] on the host CPUs, the Temporal-Mamba part: one clip, 256x256, clip_length 3, fp32, forward
    only, through the v3 tri-directional Mamba module of each of the four stages (mamba_v3_forward_ref, the pure-PyTorch
    composition of causal_conv1d_ref and selective_scan_ref); each stage is timed once and counted twice (depths
    [2,2,2,2], modeling/vivim.py:165).  SegFormer and the MLPs are not part of it."""
    import math
    nf, img, N, expand = 3, 256, 16, 2
    g = torch.Generator().manual_seed(1)
    total, per_stage = 0.0, []
    for dim, stride in zip((64, 128, 320, 512), (4, 8, 16, 32)):
        d_in, R, L = expand * dim, math.ceil(dim / 16), nf * (img // stride) ** 2
        rn = lambda *sh: torch.randn(*sh, generator=g)
        p = {"in_proj.weight": rn(2 * d_in, dim) / math.sqrt(dim), "out_proj.weight": rn(dim, d_in) / math.sqrt(d_in)}
        for sfx in ("", "_b", "_s"):
            p[f"conv1d{sfx}.weight"], p[f"conv1d{sfx}.bias"] = rn(d_in, 1, 4) / 2, rn(d_in) / 10
            p[f"x_proj{sfx}.weight"] = rn(R + 2 * N, d_in) / math.sqrt(d_in)
            p[f"dt_proj{sfx}.weight"], p[f"dt_proj{sfx}.bias"] = rn(d_in, R) / math.sqrt(R), torch.full((d_in,), -4.0)
            p[f"A{sfx}_log"] = torch.log(torch.arange(1, N + 1, dtype=torch.float32)).repeat(d_in, 1)
            p[f"D{sfx}"] = torch.ones(d_in)
        x = rn(1, L, dim)
        t0 = time.perf_counter()
        with torch.no_grad():
            y = ref_torch.mamba_v3_forward_ref(x, p, nf)
        dt = time.perf_counter() - t0
        assert torch.isfinite(y).all()
        per_stage.append(round(dt, 3))
        total += 2 * dt
    return {"seconds_per_clip": round(total, 3), "frames_per_s": round(nf / total, 3), "stage_seconds": per_stage,
            "what": "8 v3 Mamba modules (4 stages x 2) of one 256x256 clip of 3 frames, fp32 forward, pure-PyTorch reference port"}


def measured_copy_ceiling(dev, nbytes=1 << 30, iters=10):
    """Device-to-device copy rate on this GPU (read + write bytes / time): the practical HBM ceiling SURVEY.md 8d asks to
    report beside the nominal 8 TB/s.  Runs after the timed region."""
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev).fill_(1)
    dst = torch.empty_like(src)
    for _ in range(2):
        dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize(dev)
    return round(2 * nbytes * iters / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)


def pmc_record(entry_point):
    """What rocprofv3 --pmc passes over this very command measured per launch of a hot-path entry point (bench.py cannot run
    the profiler on itself): HBM bytes (FETCH_SIZE / WRITE_SIZE) and VALU wave-instructions (SQ_INSTS_VALU), from the newest
    committed profiles/r*_bench_pmc_traffic.json (tools/pmc_bench_traffic.py).  The record carries a hash of the kernel
    sources it was taken on: a record of other kernels is reported as stale, not as a measurement.
    -> (hbm_bytes | None, valu_insts | None, source file | None, stale: bool)"""
    import glob
    import re
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_pmc_traffic.json")),
                   key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])     # r01_v10 after r01_v9
    if not files:
        return None, None, None, False
    try:
        doc = json.load(open(files[-1]))
        rec = doc["per_entry_point"][entry_point]
        from pmc_bench_traffic import kernels_sha
        stale = doc.get("kernels_sha") != kernels_sha()
        return (int(rec["hbm_bytes_per_launch"]), rec.get("valu_wave_insts_per_launch"), os.path.relpath(files[-1], ROOT), stale)
    except (KeyError, ValueError, OSError, ImportError):
        return None, None, None, False


VALU_PEAK_GINST = 1024 * 2.4 / 2      # G wave-instructions per s: 1024 SIMD-32 x 2.4 GHz, one wave64 instruction per 2 cycles

# The scans' OTHER roof, spelled out.  A state update (one token, one channel, one state) needs at
# least these vector instructions; a wave64 instruction does 64 of them.  Issue time per instruction class, measured on
# MI355X at >= 2 waves per SIMD (profiles/r02_valu_lab2.log, r02_valu_lab3.log; tools/valu_lab_gen.py): plain fp32
# fma / mul / add on VGPRs 1.3 ns per wave-instruction and SIMD, v_exp_f32 3.5 ns, DPP / cross-lane forms 1.9 ns.
VALU_NS = {"plain": 1.3, "exp": 3.5, "dpp": 1.9}
VALU_IDEAL = {   # instruction classes per state update
    # h = exp2(delta A) h + (delta u) B; y += h C: mul, exp, mul, fma, fma
    "selective_scan_fwd": {"plain": 4, "exp": 1, "dpp": 0},
    # + the segment pre-pass when the token axis is cut for parallelism (every shape here): mul, exp, mul, fma
    "selective_scan_fwd_split": {"plain": 7, "exp": 2, "dpp": 0},
    # forward states again (mul, exp, mul, fma) + g, a g, x = a g h, g B, A x, dA, dB, dC (8) + the two sums over the states,
    # which no mapping gets without cross-lane work (>= 4 DPP-class instructions per update in ours: DESIGN.md 4.10)
    "selective_scan_bwd": {"plain": 11, "exp": 1, "dpp": 4},
    # + the closed-form pre-pass of the token-axis cut: mul, exp, fma
    "selective_scan_bwd_split": {"plain": 13, "exp": 2, "dpp": 4},
}
N_SIMD = 1024


def valu_ceiling(kernel, state_updates, alg_bytes):
    """-> dict: the least VALU issue time of one launch (all 1024 SIMDs busy, nothing but the instructions above) and the
    algorithmic GB/s that time allows, capped by the HBM peak."""
    mix = VALU_IDEAL[kernel + "_split"]
    ns_per_wave_update = sum(mix[c] * VALU_NS[c] for c in mix)
    t = state_updates / 64.0 / N_SIMD * ns_per_wave_update * 1e-9
    return {"instr_per_update": mix, "ns_per_wave_update": round(ns_per_wave_update, 2), "ideal_valu_us": round(t * 1e6, 1),
            "ceiling_GBps": round(min(HBM_PEAK_GBS, alg_bytes / t / 1e9), 1)}


def kbench_traffic():
    """-> {cfg: {entry point: record}} from the newest committed profiles/r*_kbench_pmc_traffic.json (tools/pmc_kbench.sh), or
    {} when none is committed or it was taken on other kernel sources."""
    import glob
    import re
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kbench_pmc_traffic.json")),
                   key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])
    if not files:
        return {}, None, False
    try:
        from pmc_bench_traffic import kernels_sha
        doc = json.load(open(files[-1]))
        return doc["per_config"], os.path.relpath(files[-1], ROOT), doc.get("kernels_sha") != kernels_sha()
    except (KeyError, ValueError, OSError, ImportError):
        return {}, None, False


def roofline_by_config(dev, iters=4):
    """The scan kernels alone at the stage-0 shapes of BASELINE.json configs[1], [2] and [4] (the grouped v3 layout: three
    directions side by side, dim = 3 * d_inner, n_groups = 3), a few launches each through the C ABI, HIP-event timed like the
    timed region: configs[2] is BASELINE's HBM-bandwidth headline, configs[4] its wide-state stress.  Runs after the timed
    region; rank 0 only."""
    import selective_scan_cuda as ss
    from vivim_amd import _lib
    out = {}
    traffic, traffic_src, traffic_stale = kbench_traffic()
    for key, (B, nf, img, N, expand, dt) in {"cfg2": (3, 5, 256, 16, 2, torch.bfloat16), "cfg3": (8, 5, 512, 16, 2, torch.float32),
                                             "cfg5": (1, 8, 256, 64, 4, torch.bfloat16)}.items():
        G, D, L = 3, 64 * expand * 3, nf * (img // 4) ** 2
        g = torch.Generator(device=dev).manual_seed(7)
        mk = lambda *sh: torch.randn(*sh, device=dev, generator=g).to(dt)
        u, z, dout = mk(B, D, L), mk(B, D, L), mk(B, D, L)
        delta = (0.2 * torch.randn(B, D, L, device=dev, generator=g)).to(dt)
        A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(D, 1)
        Bm, Cm = mk(B, G, N, L), mk(B, G, N, L)
        Dv, bias = torch.ones(D, device=dev), torch.full((D,), -4.0, device=dev)
        res = ss.fwd(u, delta, A, Bm, Cm, Dv, z, bias, True)          # warm-up + the checkpoints the backward reads
        dz = torch.empty_like(z)
        ss.bwd(u, delta, A, Bm, Cm, Dv, z, bias, dout, res[1], res[0], dz, True, False)
        _lib.profile_begin()
        for _ in range(iters):
            r = ss.fwd(u, delta, A, Bm, Cm, Dv, z, bias, True)
            ss.bwd(u, delta, A, Bm, Cm, Dv, z, bias, dout, r[1], r[0], dz, True, False)
        rec = _lib.profile_end()
        e = {"shape": {"batch": B, "dim": D, "n_groups": G, "dstate": N, "seqlen": L, "dtype": str(dt).replace("torch.", "")},
             "checkpoint_row_tokens": L // res[1].shape[2] if res[1].shape[2] else L}
        for name in ("vivim_selective_scan_fwd", "vivim_selective_scan_bwd"):
            rows = [x for x in rec if x[0] == name]
            sec, nb = sum(x[2] for x in rows), sum(x[1] for x in rows)
            short = name.replace("vivim_", "")
            ceil = valu_ceiling(short, B * D * L * N, nb / len(rows))
            tr = traffic.get(key, {}).get(short, {})
            e[short] = {"avg_us": round(sec / len(rows) * 1e6, 1), "achieved_GBps": round(nb / sec / 1e9, 1),
                        "frac": round(nb / sec / 1e9 / HBM_PEAK_GBS, 4),
                        "state_updates_per_launch": B * D * L * N,
                        "algorithmic_bytes_per_launch": int(nb / len(rows)),
                        # the lower of the two roofs for THIS shape, and how far the launch is from it
                        "ceiling_GBps": ceil["ceiling_GBps"], "frac_of_ceiling": round(nb / sec / 1e9 / ceil["ceiling_GBps"], 4),
                        "valu_ideal": ceil,
                        "traffic": None if traffic_stale else tr.get("hbm_bytes_per_launch"),
                        "traffic_source": traffic_src, "traffic_stale": bool(traffic_stale),
                        "valu_wave_insts_per_launch": None if traffic_stale else tr.get("valu_wave_insts_per_launch")}
        out[key] = e
        del u, z, dout, delta, Bm, Cm, res, dz, r
        torch.cuda.empty_cache()
    return out


def main():
    a = parse()
    # stdout carries the ONE result line and nothing else: RCCL prints a version banner to fd 1 when the first
    # communicator comes up, other libraries may do the same.  Everything else goes to stderr.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    from vivim_amd import _lib, dp
    from vivim_amd.train_step import build_model, make_optimizer, synthetic_batch, train_step
    world, rank, local_rank = dp.dist_env()
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    # VIVIM_DP_BACKEND=gloo (rehearsal only): several ranks on the GPUs there are -- on a one-GPU box every rank lands on cuda:0
    # and the gradients are reduced through the host: the whole N > 1 code path (DDP buckets over real device tensors, fused
    # parameter views, max-over-ranks timing) without an 8-GPU node.  The driver's runs use RCCL.
    backend = os.environ.get("VIVIM_DP_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dp.init(backend, dev)                                      # RCCL over xGMI when WORLD_SIZE > 1
    amp = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[a.dtype]

    _lib.lib()                                                 # fail loudly if the HIP library is missing

    torch.manual_seed(a.seed)                                  # identical replicas
    model = build_model(a.num_classes, dev, mamba_kwargs={"d_state": a.d_state, "expand": a.expand},
                        fast_backbone_dwconv=not a.stock_backbone_dwconv)
    clip, onehot = synthetic_batch(a.train_bs, a.clip_length, a.image_size, a.num_classes, dev,
                                   dp.shard_seed(a.seed, rank))

    def barrier():
        dp.barrier(dev)

    # Eager step; DDP 25 MB buckets in reverse registration order so the stage-3/2 buckets fly while the long
    # stage-0/1 backward scans still run (vivim_amd/dp.py; identity at N = 1).  (Replaying the whole step from a
    # HIP graph was tried: MIOpen / hipBLASLt gradient kernels of the SegFormer blocks are not replay-stable on
    # this ROCm build -- non-finite bias / sr-conv gradients from the second replay on, tools/graph_probe.py.)
    step_model = dp.wrap(model, dev)
    opt = make_optimizer(model)
    mode = "eager"

    for _ in range(a.warmup):
        train_step(step_model, opt, clip, onehot, a.num_classes, amp)
    barrier()
    _lib.profile_begin()                     # HIP events around the hot-path kernels only (pooled events)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = train_step(step_model, opt, clip, onehot, a.num_classes, amp)
    barrier()
    elapsed = time.perf_counter() - t0
    records = _lib.profile_end()
    elapsed = dp.max_over_ranks(elapsed, dev)
    assert torch.isfinite(loss), "non-finite loss"
    comm = {}
    try:                                                       # every rank takes part; rank 0 reports
        comm = dp.comm_probe(step_model, lambda: train_step(step_model, opt, clip, onehot, a.num_classes, amp),
                             elapsed / a.steps * 1e3, dev)
    except Exception as ex:
        comm = {"error": repr(ex)[:200]}

    if rank == 0:
        frames = world * a.train_bs * a.clip_length * a.steps
        per = {}
        for name, nbytes, sec in records:
            e = per.setdefault(name, [0, 0.0, 0])
            e[0] += nbytes
            e[1] += sec
            e[2] += 1
        hot = [k for k in per if "selective_scan" in k or "causal_conv1d" in k]   # the north-star path's kernels
        dom = max(hot, key=lambda k: per[k][1])
        ach = per[dom][0] / per[dom][1] / 1e9
        traffic, valu_insts, traffic_src, stale = pmc_record(dom.replace("vivim_", ""))
        avg_s = per[dom][1] / per[dom][2]
        # the VALU co-limit: the scans carry N exp2 + ~20 N fp32 operations per token and channel; their issue rate against
        # the chip's (one wave64 instruction per 2 cycles and SIMD) says how far the kernel is from its OTHER roof
        co = None
        if valu_insts and not stale:
            co = {"kind": "valu", "wave_insts_per_launch": int(valu_insts),
                  "achieved_Ginst_per_s": round(valu_insts / avg_s / 1e9, 1), "peak_Ginst_per_s": round(VALU_PEAK_GINST, 1),
                  "frac": round(valu_insts / avg_s / 1e9 / VALU_PEAK_GINST, 4),
                  "note": "SQ_INSTS_VALU per launch (committed PMC pass) / live launch time; the larger of roofline.frac and "
                          "this one names the bound"}
        # the same ceiling for the dominant kernel over the launches of the timed region: state updates of the four grouped stage
        # shapes (two blocks each per step), from the model's sizes
        dom_short = dom.replace("vivim_", "")
        ceiling = None
        if dom_short in VALU_IDEAL:
            upd = sum(2 * a.train_bs * 3 * (a.expand * d) * (a.clip_length * (a.image_size // st) ** 2) * a.d_state
                      for d, st in zip((64, 128, 320, 512), (4, 8, 16, 32)))              # per step and rank
            c = valu_ceiling(dom_short, upd / 8.0, per[dom][0] / per[dom][2])
            ceiling = {"ceiling_GBps": c["ceiling_GBps"], "frac_of_ceiling": round(ach / c["ceiling_GBps"], 4),
                       "valu_ideal_us_per_launch": c["ideal_valu_us"], "instr_per_update": c["instr_per_update"],
                       "note": "min(HBM peak, algorithmic bytes / least VALU issue time): DESIGN.md 4.5"}
        kernels = {k.replace("vivim_", ""): {"launches": v[2], "total_ms": round(v[1] * 1e3, 3),
                                              "avg_us": round(v[1] / v[2] * 1e6, 2),
                                              "alg_GBps": round(v[0] / v[1] / 1e9, 1)} for k, v in per.items()}
        out = {
            "metric": f"frames/sec fwd+bwd, {a.image_size}x{a.image_size} clip={a.clip_length} {a.num_classes}-class",
            "value": round(frames / elapsed, 3),
            "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "Vivim train step (fwd+loss+bwd+AdamW), BASELINE.json configs[1]"
                       if (a.image_size, a.clip_length, a.train_bs, a.dtype) == (256, 5, 3, "bf16")
                       else "Vivim train step (fwd+loss+bwd+AdamW), non-default sizes",
                       "image_size": a.image_size, "clip_length": a.clip_length, "num_classes": a.num_classes,
                       "per_gpu_batch": a.train_bs, "global_batch": a.train_bs * world, "d_state": a.d_state,
                       "backbone": "SegFormer-b3 architecture, random init", "parallelism": f"dp{world}",
                       "step_mode": mode},
            "roofline": {"bound": "hbm", "kernel": dom.replace("vivim_", ""), "achieved": round(ach, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                         "traffic": None if stale else traffic, "traffic_unit": "bytes per launch (PMC, separate passes)",
                         "traffic_source": traffic_src, "traffic_stale": bool(stale), "co_limit": co, "valu_ceiling": ceiling,
                         "algorithmic_bytes_per_launch": int(per[dom][0] / per[dom][2]),
                         "launches": per[dom][2],
                         "avg_launch_us": round(per[dom][1] / per[dom][2] * 1e6, 2),
                         "measured_copy_GBps": measured_copy_ceiling(dev)},
            "kernels": kernels,
            # what the step holds in HBM (caching allocator peak since process start; the lanes = states family keeps 4 bytes of
            # scan checkpoints per token and channel from forward to backward -- ADVICE round 2: compare VIVIM_FWD_VARIANT=1)
            "max_memory_allocated_MB": round(torch.cuda.max_memory_allocated(dev) / 1e6, 1),
            "loss": round(float(loss), 5),
        }
        if comm:
            out["comm"] = comm                                 # allreduce_ms, overlap_frac, n_ranks_seen (N > 1)
        if world == 1 and not a.no_by_config:
            try:
                out["roofline_by_config"] = roofline_by_config(dev)
            except Exception as ex:                            # never lose the headline line to the extras
                out["roofline_by_config"] = {"error": repr(ex)[:200]}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
