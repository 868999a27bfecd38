#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc SQ pass per kernel: mean counter value per dispatch and the wave-cycle split.
Usage: python tools/pmc_sq.py <dir> [name-substring ...]
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md)."""
import csv, glob, os, sys
from collections import defaultdict
out = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
subs = sys.argv[2:] or ["vivim"]
for k in sorted(out):
    if not any(s in k for s in subs):
        continue
    m = {c: sum(v) / len(v) for c, v in out[k].items()}
    print(k[:100], "dispatches", max(len(v) for v in out[k].values()))
    wc = m.get("SQ_WAVE_CYCLES")
    for c in sorted(m):
        extra = f"  ({100 * m[c] / wc:5.1f}% of wave cycles)" if wc and c.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_INST_CYCLES")) else ""
        print(f"   {c:28s} {m[c]:16.0f}{extra}")
    if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m:
        print(f"   VALU instr per wave          {m['SQ_INSTS_VALU'] / m['SQ_WAVES']:16.1f}")
