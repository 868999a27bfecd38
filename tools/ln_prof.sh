#!/bin/bash
# rocprofv3 kernel durations of tools/ln_bench.py, one stage shape per run:  bash tools/ln_prof.sh <tag>
set -e -o pipefail
tag=${1:?tag}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for s in 0 1 2 3; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/lnprof_${tag}_$s" -o k -- python3 "$root/tools/ln_bench.py" $s > "$out/lnprof_${tag}_$s.log" 2>&1
  f=$(find "$out/lnprof_${tag}_$s" -name "*kernel_stats.csv" | head -n 1)
  echo "== stage $s: $(grep stage "$out/lnprof_${tag}_$s.log")"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "ln_" in n or "layer_norm" in n or "GammaBeta" in n or "GradInput" in n:
        print("   %-70s calls %4s avg %8.1f us" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf "$out/lnprof_${tag}_$s"
done
