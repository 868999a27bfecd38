#!/bin/bash
# SQ instruction-mix pass (dynamic SALU / SMEM / LDS / VMEM counts beside VALU) over one tools/kbench.py invocation:  bash tools/kpmc.sh <tag> <kbench args...>
# (counters in their own run, kernel trace only -- never combined with the hip/hsa trace domains)
set -e -o pipefail
tag=${1:?tag}; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA \
    --output-format csv -d "$out/kpmi_$tag" -o k -- python3 "$root/tools/kbench.py" "$@" > "$out/kpmi_$tag.log" 2>&1
cd "$root"
python tools/pmc_sq.py "$out/kpmi_$tag" ssm_ls ssm_bwd ssm_fwd > "$out/kpmi_$tag.txt"
rm -rf "$out/kpmi_$tag"
cat "$out/kpmi_$tag.txt"
