#!/bin/bash
# FETCH_SIZE calibration on known byte counts (tools/fetch_cal.hip):  gpurun -- 'bash tools/fetch_cal.sh'
set -e -o pipefail
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/fcal_t" -o f -- "$root/tools/fetch_cal" > "$out/fcal.log" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fcal_1" -o f -- "$root/tools/fetch_cal" > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d "$out/fcal_2" -o f -- "$root/tools/fetch_cal" > /dev/null 2>&1
cd "$root"
python - "$out" <<'PY' > "$out/fetch_cal.txt"
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for d in ("fcal_1", "fcal_2"):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(os.path.join(out, "fcal_t", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = float(r["AverageNs"]) / 1e3
GiB = float(1 << 30)
for k in sorted(acc):
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    nbytes = GiB if "wide" in k else (26214 - 26214 % (16 if "piece" in k else 64)) * 40960.0
    print(f"{k[:60]:60s} read once: {nbytes / 1e6:8.1f} MB   FETCH_SIZE {m.get('FETCH_SIZE', 0) * 1024 / 1e6:8.1f} MB "
          f"(x{nbytes / max(m.get('FETCH_SIZE', 0) * 1024, 1):.2f} to the truth)   RDREQ {m.get('TCC_EA0_RDREQ_sum', 0):.0f} "
          f"(32 B: {m.get('TCC_EA0_RDREQ_32B_sum', 0):.0f}) = {nbytes / max(m.get('TCC_EA0_RDREQ_sum', 0), 1):.1f} B per request   "
          f"{dur.get(k, 0):8.1f} us = {nbytes / max(dur.get(k, 1), 1) / 1e6:.2f} TB/s")
PY
rm -rf "$out/fcal_1" "$out/fcal_2" "$out/fcal_t"
cat "$out/fetch_cal.txt"
