#!/bin/bash
# HBM traffic + VALU instruction counts of the grouped stage-0 scans of BASELINE configs 2 / 3 / 5 (what bench.py's
# roofline_by_config times), three PMC passes per config over tools/kbench.py -> <tag>_kbench_pmc_traffic.json
#   gpurun --timeout 900 -- 'bash tools/pmc_kbench.sh r03_v1'      then copy gpurun_out/<tag>_kbench_pmc_traffic.json to profiles/
set -e -o pipefail
tag=${1:?tag}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
scratch=$out/pmck_$tag
mkdir -p "$scratch"
cd /tmp && export TMPDIR=/tmp
for cfg in 2 3 5; do
  it=3
  i=0
  for c in FETCH_SIZE "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" SQ_INSTS_VALU; do
    name=$(echo fetch write valu | cut -d' ' -f$((i+1))); i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$scratch/cfg${cfg}_$name" -o k -- python3 "$root/tools/kbench.py" --config $cfg --groups 3 --stages 0 --kernels sf,sb --iters $it > "$scratch/cfg${cfg}_$name.log" 2>&1 || echo "cfg $cfg pass $name failed"
  done
  echo "cfg $cfg done"
done
cd "$root"
python tools/pmc_bench_traffic.py --kbench "$scratch" "$out/${tag}_kbench_pmc_traffic.json"
rm -rf "$scratch"
