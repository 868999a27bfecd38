#!/usr/bin/env python3
"""HBM traffic of the benchmark's hot-path kernels from two rocprofv3 --pmc passes over bench.py (FETCH_SIZE and
WRITE_SIZE cannot share a pass on gfx950; FETCH_SIZE reports half of a wide coalesced read stream there -- see
MI355X_MICROARCH.md, HBM section -- so reads are doubled).  Kernels are grouped by the C-ABI entry point that launches
them; the result is bytes per entry-point launch, like roofline.achieved in bench.py.
Usage: python tools/pmc_bench_traffic.py <dir-FETCH_SIZE-pass> <dir-WRITE_SIZE-pass> <launches-per-entry-point-json> <out.json>"""
import csv, glob, json, os, sys
from collections import defaultdict

GROUPS = {"selective_scan_bwd": ("ssm_bwd",), "selective_scan_fwd": ("ssm_fwd",),
          "causal_conv1d_fwd": ("conv1d_fwd",), "causal_conv1d_bwd": ("conv1d_bwd",)}


def total(d, counter):
    out = defaultdict(float)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                for g, subs in GROUPS.items():
                    if any(s in r["Kernel_Name"] for s in subs):
                        out[g] += float(r["Counter_Value"])
    return out


fetch, write = total(sys.argv[1], "FETCH_SIZE"), total(sys.argv[2], "WRITE_SIZE")
launches = json.load(open(sys.argv[3]))          # {"selective_scan_bwd": launches in the profiled run incl. warm-up, ...}
res = {}
for g in GROUPS:
    n = launches[g]
    res[g] = {"launches_profiled": n, "fetch_KiB_per_launch": round(fetch[g] / n, 1), "write_KiB_per_launch": round(write[g] / n, 1),
              "hbm_bytes_per_launch": int((2 * fetch[g] + write[g]) * 1024 / n)}
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `bench.py --steps 5 --warmup 2`; reads doubled (gfx950)",
           "per_entry_point": res}, open(sys.argv[4], "w"), indent=1)
print(json.dumps(res, indent=1))
