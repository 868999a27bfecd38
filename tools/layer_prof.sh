#!/bin/bash
# GPU time of one MambaLayer forward + backward, kernel by kernel, per stage shape of configs[1]:  bash tools/layer_prof.sh <tag>
set -e -o pipefail
tag=${1:?tag}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for s in 0 1 2 3; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/layer_${tag}_$s" -o k -- python3 "$root/tools/layer_prof.py" $s 10 > "$out/layer_${tag}_$s.log" 2>&1
  f=$(find "$out/layer_${tag}_$s" -name "*kernel_stats.csv" | head -n 1)
  python3 - "$f" "$s" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 12 / 1e3
ours = sum(float(r["TotalDurationNs"]) for r in rows if "vivim" in r["Name"]) / 12 / 1e3
print("== stage %s: %.0f us of GPU time per MambaLayer forward + backward, %.0f us of it in this library's kernels" % (sys.argv[2], tot, ours))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:28]:
    print("   %-96s x%-4.1f avg %8.1f us   %6.1f us per iteration" % (r["Name"][:96], int(r["Calls"]) / 12, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 12 / 1e3))
PY
  rm -rf "$out/layer_${tag}_$s"
done
