#!/bin/bash
# HBM-side counters of one tools/kbench.py invocation, per kernel:  bash tools/khbm.sh <tag> <kbench args...>
# Three passes (FETCH_SIZE and WRITE_SIZE cannot share one; the raw request counters calibrate FETCH_SIZE for access widths
# other than 16 B per lane, MI355X_MICROARCH.md HBM section).  Kernel trace only.
set -e -o pipefail
tag=${1:?tag}; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/khbm_${tag}_$i" -o k -- python3 "$root/tools/kbench.py" "$@" > "$out/khbm_${tag}_$i.log" 2>&1 || echo "pass $i ($c) failed"
done
cd "$root"
python tools/pmc_sq.py "$out" ssm_ls ssm_bwd ssm_fwd conv1d > "$out/khbm_$tag.txt" 2>&1 || true
python - "$out" "$tag" <<'PY' > "$out/khbm_$tag.txt"
import csv, glob, os, sys
from collections import defaultdict
out, tag = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for d in glob.glob(os.path.join(out, f"khbm_{tag}_*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "vivim" in r["Kernel_Name"]:
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    print(k[:110], "dispatches", max(len(v) for v in acc[k].values()))
    for c in sorted(m):
        print(f"   {c:28s} {m[c]:16.1f}")
    if "TCC_EA0_RDREQ_sum" in m:
        r32 = m.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        print(f"   read requests: {m['TCC_EA0_RDREQ_sum']:.0f} of which 32 B {r32:.0f}; bytes if the rest are 64 B: "
              f"{(r32 * 32 + (m['TCC_EA0_RDREQ_sum'] - r32) * 64) / 1e6:.1f} MB, if 128 B: {(r32 * 32 + (m['TCC_EA0_RDREQ_sum'] - r32) * 128) / 1e6:.1f} MB")
    if "FETCH_SIZE" in m:
        print(f"   FETCH_SIZE {m['FETCH_SIZE'] * 1024 / 1e6:.1f} MB (x2 = {2 * m['FETCH_SIZE'] * 1024 / 1e6:.1f} MB)   WRITE_SIZE {m.get('WRITE_SIZE', 0) * 1024 / 1e6:.1f} MB")
PY
rm -rf "$out"/khbm_${tag}_[0-9]
cat "$out/khbm_$tag.txt"
