#!/bin/bash
# SQ counter pass over one tools/kbench.py invocation:  bash tools/kpmc.sh <tag> <kbench args...>
# (counters in their own run, kernel trace only -- never combined with the hip/hsa trace domains)
set -e -o pipefail
tag=${1:?tag}; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    --output-format csv -d "$out/kpmc_$tag" -o k -- python3 "$root/tools/kbench.py" "$@" > "$out/kpmc_$tag.log" 2>&1
cd "$root"
python tools/pmc_sq.py "$out/kpmc_$tag" ssm_ls ssm_bwd ssm_fwd > "$out/kpmc_$tag.txt"
rm -rf "$out/kpmc_$tag"
cat "$out/kpmc_$tag.txt"
