#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` directory into a small, committable
summary: top-N kernels by total time with calls / total / average / percentage (names truncated).
Usage: python tools/prof_summary.py gpurun_out/prof1 profiles/r01_xxx.csv [N]"""
import csv
import glob
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
files = glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True)
assert files, "no *_kernel_stats.csv under " + src
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows[:top]:
        w.writerow([r["Name"][:140], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                    r["MinNs"], r["MaxNs"]])
print("wrote", dst, "from", files[0])
