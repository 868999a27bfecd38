#!/usr/bin/env python3
"""HBM traffic and VALU instruction counts of the benchmark's hot-path kernels from rocprofv3 --pmc passes over bench.py
(FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950; FETCH_SIZE reports half of a wide coalesced read stream there --
MI355X_MICROARCH.md, HBM section -- so reads are doubled).  Kernels are grouped by the C-ABI entry point that launches
them; launches are counted from the profile itself (dispatches of the entry point's main kernel); the record carries a
hash of the kernel sources so that bench.py can tell a stale record from a current one.
Usage: python tools/pmc_bench_traffic.py <dir-FETCH_SIZE-pass> <dir-WRITE_SIZE-pass> <dir-SQ_INSTS_VALU-pass | -> <out.json>"""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = {"selective_scan_bwd": ("ssm_bwd", "ssm_ls_bwd", "ssm_ls2_bwd", "ssm_ls_carry_kernelILb1", "ssm_ls_carry_kernel<true>"),
          "selective_scan_fwd": ("ssm_fwd", "ssm_ls_fwd", "ssm_ls_carry_kernelILb0", "ssm_ls_carry_kernel<false>"),
          "causal_conv1d_fwd": ("conv1d_fwd",), "causal_conv1d_bwd": ("conv1d_bwd",)}
# one dispatch of these per entry-point call
MAIN = {"selective_scan_bwd": ("ssm_ls_bwd_kernel", "ssm_ls2_bwd_kernel", "ssm_bwd_fast_kernel", "ssm_bwd_generic_kernel"),
        "selective_scan_fwd": ("ssm_fwd_nsplit_kernel", "ssm_fwd_generic_kernel",
                               "ssm_fwd_bc_kernel",            # lanes=channels: once per call (its two passes share a name)
                               "ELi2ELb", ", 2, true>", ", 2, false>"),       # lanes=states: PASS == 2
        "causal_conv1d_fwd": ("conv1d_fwd",), "causal_conv1d_bwd": ("conv1d_bwd",)}


def kernels_sha():
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "vivim_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "vivim_amd", "csrc", "*.cuh"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def total(d, counter):
    out, main = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for g, subs in GROUPS.items():
                if any(s in r["Kernel_Name"] for s in subs):
                    out[g] += float(r["Counter_Value"])
                    if any(s in r["Kernel_Name"] for s in MAIN[g]):
                        main[g] += 1
    return out, main


def read_kib(fetch_dir):
    """FETCH_SIZE in KiB per entry point; from the raw request counters when the derived metric was not collected (the
    derived-metric passes over bench.py die inside the profiler every other run; FETCH_SIZE = 64 B x requests, none of them
    32-byte ones, in every kernel here: profiles/r02_fetch_size_calibration.txt)."""
    fetch, nf = total(fetch_dir, "FETCH_SIZE")
    if not nf:
        (rd, nf), (rd32, _) = total(fetch_dir, "TCC_EA0_RDREQ_sum"), total(fetch_dir, "TCC_EA0_RDREQ_32B_sum")
        fetch = {g: ((rd[g] - rd32.get(g, 0.0)) * 64 + rd32.get(g, 0.0) * 32) / 1024 for g in rd}
    return fetch, nf


def record(fetch_dir, write_dir, valu_dir):
    (fetch, nf) = read_kib(fetch_dir)
    (wr, nw), (wr64, _) = total(write_dir, "TCC_EA0_WRREQ_sum"), total(write_dir, "TCC_EA0_WRREQ_64B_sum")
    write = {g: ((wr[g] - wr64.get(g, 0.0)) * 32 + wr64.get(g, 0.0) * 64) / 1024 for g in wr}
    valu, nv = total(valu_dir, "SQ_INSTS_VALU") if valu_dir != "-" else ({}, {})
    res = {}
    for g in GROUPS:
        if not nf.get(g) or not nw.get(g):
            continue
        res[g] = {"launches_profiled": nf[g], "fetch_KiB_per_launch": round(fetch[g] / nf[g], 1),
                  "write_KiB_per_launch": round(write[g] / nw[g], 1),
                  "hbm_bytes_per_launch": int((2 * fetch[g] / nf[g] + write[g] / nw[g]) * 1024)}
        if nv.get(g):
            res[g]["valu_wave_insts_per_launch"] = int(valu[g] / nv[g])
    return res


if __name__ == "__main__" and sys.argv[1] == "--kbench":
    # python tools/pmc_bench_traffic.py --kbench <scratch dir with cfgN_{fetch,write,valu} pass directories> <out.json>
    doc = {"note": "rocprofv3 --pmc passes (one counter set per pass) over `tools/kbench.py --config N --groups 3 --stages 0 --kernels "
                   "sf,sb`: the grouped stage-0 scans of BASELINE configs 2 / 3 / 5; reads doubled (gfx950)",
           "kernels_sha": kernels_sha(), "per_config": {}}
    for cfg in ("cfg2", "cfg3", "cfg5"):
        d = os.path.join(sys.argv[2], cfg)
        if os.path.isdir(d + "_fetch"):
            doc["per_config"][cfg] = record(d + "_fetch", d + "_write", d + "_valu" if os.path.isdir(d + "_valu") else "-")
    json.dump(doc, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(doc["per_config"], indent=1))
elif __name__ == "__main__" and sys.argv[1] == "--mix":
    # python tools/pmc_bench_traffic.py --mix <scratch dir with mix_{fetch,write,valu} pass directories> <out.json>
    # The launch mix of the bench step (the four grouped stage shapes of configs[1], every entry point once per stage and
    # iteration) issued by tools/kbench.py instead of bench.py: counter mode over the whole training step dies inside the
    # profiler under ATen's LayerNorm backward launch every other run (profiles/r03_pmc_*_profiler_abort.log).
    d = sys.argv[2]
    res = record(os.path.join(d, "mix_fetch"), os.path.join(d, "mix_write"),
                 os.path.join(d, "mix_valu") if os.path.isdir(os.path.join(d, "mix_valu")) else "-")
    json.dump({"note": "rocprofv3 --pmc passes (raw TCC_EA0 request counters, one set per pass; SQ_INSTS_VALU) over `tools/kbench.py "
                       "--config 2 --groups 3 --stages 0,1,2,3 --kernels sf,sb,cf,cb`: the launch shapes of `bench.py`'s step "
                       "(BASELINE configs[1]: four stages, equal weight), per launch; reads doubled (gfx950)",
               "kernels_sha": kernels_sha(), "per_entry_point": res}, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(res, indent=1))
elif __name__ == "__main__":
    (fetch, nf) = read_kib(sys.argv[1])
    (write, nw) = total(sys.argv[2], "WRITE_SIZE")
    if not nw:                                       # raw request counters instead of the derived metric (KiB like WRITE_SIZE)
        (wr, nw), (wr64, _) = total(sys.argv[2], "TCC_EA0_WRREQ_sum"), total(sys.argv[2], "TCC_EA0_WRREQ_64B_sum")
        write = {g: ((wr[g] - wr64.get(g, 0.0)) * 32 + wr64.get(g, 0.0) * 64) / 1024 for g in wr}
    valu, nv = total(sys.argv[3], "SQ_INSTS_VALU") if sys.argv[3] != "-" else ({}, {})
    res = {}
    for g in GROUPS:
        if not nf.get(g) or not nw.get(g):
            continue
        res[g] = {"launches_profiled": nf[g], "fetch_KiB_per_launch": round(fetch[g] / nf[g], 1),
                  "write_KiB_per_launch": round(write[g] / nw[g], 1),
                  "hbm_bytes_per_launch": int((2 * fetch[g] / nf[g] + write[g] / nw[g]) * 1024)}
        if nv.get(g):
            res[g]["valu_wave_insts_per_launch"] = int(valu[g] / nv[g])
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU passes (one counter per pass) over `bench.py --steps 2 "
                       "--warmup 1`; reads doubled (gfx950); launches counted from the profile",
               "kernels_sha": kernels_sha(), "per_entry_point": res}, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(res, indent=1))
