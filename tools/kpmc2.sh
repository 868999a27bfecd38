#!/bin/bash
# second SQ counter pass (LDS / memory side) over one tools/kbench.py invocation:  bash tools/kpmc2.sh <tag> <kbench args...>
set -e -o pipefail
tag=${1:?tag}; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS \
    --output-format csv -d "$out/kpmc2_$tag" -o k -- python3 "$root/tools/kbench.py" "$@" > "$out/kpmc2_$tag.log" 2>&1
cd "$root"
python tools/pmc_sq.py "$out/kpmc2_$tag" ssm_ls ssm_bwd ssm_fwd > "$out/kpmc2_$tag.txt"
rm -rf "$out/kpmc2_$tag"
cat "$out/kpmc2_$tag.txt"
