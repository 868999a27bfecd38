#!/bin/bash
# kernel durations of tools/wgrad_bench.py per stage shape:  bash tools/wgrad_prof.sh   (VIVIM_WGRAD_* are inherited)
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for s in 0 1 2 3; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/wgp_$s" -o k -- python3 "$root/tools/wgrad_bench.py" $s > "$out/wgp_$s.log" 2>&1
  f=$(find "$out/wgp_$s" -name "*kernel_stats.csv" | head -n 1)
  python3 - "$f" "$s" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wgrad_nt" in r["Name"] or r["Name"].startswith("Cijk")]
print("stage", sys.argv[2], "  ".join("%s %.1f us" % (("wgrad NF=1" if ("Li1E" in r["Name"] or ", 1>" in r["Name"]) else "wgrad NF=4") if "wgrad" in r["Name"] else "bmm " + r["Name"][52:64], float(r["AverageNs"]) / 1e3) for r in rows), flush=True)
PY
  rm -rf "$out/wgp_$s"
done
