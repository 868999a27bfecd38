#!/bin/bash
# The three PMC passes of measure_round.sh alone (one counter per pass, kernel trace only) + the traffic record.
#   gpurun --timeout 1000 -- 'bash tools/pmc_round.sh r02_v2'
set -e -o pipefail
tag=${1:?tag}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
# (both traffic passes collect RAW request counters -- FETCH_SIZE = 64 B x RDREQ here, profiles/r02_fetch_size_calibration.txt;
# the write pass: WRITE_SIZE = 32 B x (WRREQ - WRREQ_64B) + 64 B x WRREQ_64B, checked in
# profiles/r02_hbm_counters_per_kernel.txt; the derived WRITE_SIZE pass crashed inside the profiler every other run)
for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" SQ_INSTS_VALU; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_${c%% *}" -o b -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-by-config > "$out/pmc_${c%% *}.log" 2>&1
  echo "pass $c done"
done
cd "$root"
python tools/pmc_bench_traffic.py "$out/pmc_TCC_EA0_RDREQ_sum" "$out/pmc_TCC_EA0_WRREQ_sum" "$out/pmc_SQ_INSTS_VALU" "$out/${tag}_bench_pmc_traffic.json"
rm -rf "$out/pmc_TCC_EA0_RDREQ_sum" "$out/pmc_TCC_EA0_WRREQ_sum" "$out/pmc_SQ_INSTS_VALU"
