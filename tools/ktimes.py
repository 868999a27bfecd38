#!/usr/bin/env python3
"""Print the top kernels of tools/kprof.sh summaries: python tools/ktimes.py gpurun_out/kprof_<tag>_kernel_stats.csv ..."""
import csv, sys
for f in sys.argv[1:]:
    print(f)
    for r in list(csv.DictReader(open(f)))[:6]:
        if "vivim" in r["Name"]:
            print("  %-74s calls %4s avg %10.1f us" % (r["Name"][:74], r["Calls"], float(r["AverageNs"]) / 1e3))
