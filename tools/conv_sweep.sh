#!/bin/bash
# conv1d pipelined kernels: kernel durations (rocprofv3) against the wave target VIVIM_CONV_WAVES, cfg 2 and cfg 5 grouped shapes
#   gpurun -- 'bash tools/conv_sweep.sh > gpurun_out/conv_sweep.txt'
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p "$out"
for cfg in 2 5; do
  for w in 2048 4096 8192 16384 65536; do
    for s in 0 1 2; do
      cd /tmp && export TMPDIR=/tmp VIVIM_CONV_WAVES=$w
      timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/cs_$cfg$w$s" -o k -- python3 "$root/tools/kbench.py" --config $cfg --groups 3 --stages $s --kernels cf,cb --iters 30 > /dev/null 2>&1
      f=$(find "$out/cs_$cfg$w$s" -name "*kernel_stats.csv" | head -n 1)
      python3 - "$f" "$cfg" "$w" "$s" <<'PY'
import csv, sys
t = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "conv1d_fwd" in n: t["fwd"] = float(r["AverageNs"]) / 1e3
    if "conv1d_bwd" in n: t["bwd"] = float(r["AverageNs"]) / 1e3
print("cfg %s waves %7s stage %s: fwd %7.1f us  bwd %7.1f us" % (sys.argv[2], sys.argv[3], sys.argv[4], t.get("fwd", -1), t.get("bwd", -1)), flush=True)
PY
      rm -rf "$out/cs_$cfg$w$s"
    done
  done
done
