#!/bin/bash
# HBM traffic + VALU instruction counts per launch of the four hot-path entry points on the bench step's launch mix, three PMC
# passes over tools/kbench.py (all four grouped stage shapes of BASELINE configs[1]) -> <tag>_bench_pmc_traffic.json.
# (tools/pmc_round.sh profiles bench.py itself; rocprofv3's counter mode crashes inside ATen's LayerNorm backward launch there
# every other run, profiles/r03_pmc_*_profiler_abort.log.)
#   gpurun --timeout 900 -- 'bash tools/pmc_mix.sh r03_v5'      then copy gpurun_out/<tag>_bench_pmc_traffic.json to profiles/
set -e -o pipefail
tag=${1:?tag}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
scratch=$out/pmcm_$tag
mkdir -p "$scratch"
cd /tmp && export TMPDIR=/tmp
i=0
for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" SQ_INSTS_VALU; do
  name=$(echo fetch write valu | cut -d' ' -f$((i+1))); i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$scratch/mix_$name" -o k -- python3 "$root/tools/kbench.py" --config 2 --groups 3 --stages 0,1,2,3 --kernels sf,sb,cf,cb --iters 3 > "$scratch/mix_$name.log" 2>&1
  echo "pass $name done"
done
cd "$root"
python tools/pmc_bench_traffic.py --mix "$scratch" "$out/${tag}_bench_pmc_traffic.json"
rm -rf "$scratch"
