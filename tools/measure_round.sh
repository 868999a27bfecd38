#!/bin/bash
# One measurement pass on the GPU box: full bench line, kernel micro-benchmarks, rocprofv3 kernel stats of bench.py and
# the two PMC traffic passes.  Everything lands under gpurun_out/ with the given tag; copy what is to be kept to profiles/.
#   gpurun --timeout 1190 -- 'bash tools/measure_round.sh v8'
# Stops at the first failing GPU step (set -e): no GPU step is started after one that timed out.
set -e -o pipefail
tag=${1:?tag}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p "$out"
cd "$root"
timeout -k 10 300 python bench.py > "$out/bench_$tag.json" 2> "$out/bench_$tag.err"
{
  echo "# tools/kbench.py, MI355X, round 1 $tag"
  echo "== per-direction shapes (cfg 2)";        timeout -k 10 200 python tools/kbench.py --config 2 --iters 30 2>&1 | grep stage
  echo "== grouped v3 shapes (cfg 2, --groups 3)"; timeout -k 10 200 python tools/kbench.py --config 2 --groups 3 --iters 30 2>&1 | grep stage
  echo "== cfg 3 (fp32), per direction";         timeout -k 10 300 python tools/kbench.py --config 3 --iters 10 2>&1 | grep stage
  echo "== cfg 3 stage 0 grouped";               timeout -k 10 200 python tools/kbench.py --config 3 --stages 0 --groups 3 --iters 10 2>&1 | grep stage
} > "$out/kbench_$tag.log"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_$tag" -o bench -- python3 "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_f" -o b -- python3 "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_w" -o b -- python3 "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd "$root"
python tools/prof_summary.py "$out/prof_$tag" "$out/prof_${tag}_kernel_stats.csv" 60
# 7 profiled steps x 8 Temporal Mamba layers; the conv forward runs twice per layer step (forward + recompute in backward)
echo '{"selective_scan_bwd": 56, "selective_scan_fwd": 56, "causal_conv1d_fwd": 112, "causal_conv1d_bwd": 56}' > "$out/launches.json"
python tools/pmc_bench_traffic.py "$out/pmc_f" "$out/pmc_w" "$out/launches.json" "$out/bench_pmc_traffic_$tag.json" > /dev/null
rm -rf "$out/prof_$tag" "$out/pmc_f" "$out/pmc_w"
python - "$out/bench_$tag.json" <<'EOF'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["achieved"])
EOF
