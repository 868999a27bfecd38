#!/bin/bash
# One measurement pass on the GPU box: full bench line, kernel micro-benchmarks, rocprofv3 kernel stats of bench.py and
# (the three PMC passes are tools/pmc_round.sh: run it after this one, in the same call, joined with &&).  Everything lands
# under gpurun_out/ with the given tag; copy what is to be kept to profiles/.
#   gpurun --timeout 1190 -- 'bash tools/measure_round.sh r02_v1'
# Stops at the first failing GPU step (set -e): no GPU step is started after one that timed out.
set -e -o pipefail
tag=${1:?tag}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p "$out"
cd "$root"
timeout -k 10 400 python bench.py > "$out/bench_$tag.json" 2> "$out/bench_$tag.err"
# the same step with the long (256-token) checkpoint rows everywhere: peak memory and time beside the automatic choice
VIVIM_FWD_VARIANT=1 timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-by-config > "$out/bench_${tag}_longckpt.json" 2> /dev/null || true
{
  echo "# tools/kbench.py, MI355X, $tag (automatic kernel choice)"
  echo "== grouped v3 shapes (cfg 2, --groups 3)"; timeout -k 10 200 python tools/kbench.py --config 2 --groups 3 --iters 30 2>&1 | grep stage
  echo "== cfg 3 (fp32), per direction";         timeout -k 10 300 python tools/kbench.py --config 3 --iters 10 2>&1 | grep stage
  echo "== cfg 3 stage 0 grouped";               timeout -k 10 200 python tools/kbench.py --config 3 --stages 0 --groups 3 --iters 10 2>&1 | grep stage
  echo "== cfg 5 grouped";                       timeout -k 10 200 python tools/kbench.py --config 5 --groups 3 --iters 10 2>&1 | grep stage
} > "$out/kbench_$tag.log"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_$tag" -o bench -- python3 "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-by-config > /dev/null 2>&1
cd "$root"
python tools/prof_summary.py "$out/prof_$tag" "$out/${tag}_bench_kernel_stats.csv" 60
rm -rf "$out/prof_$tag"
python - "$out/bench_$tag.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["achieved"])
PY
