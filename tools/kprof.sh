#!/bin/bash
# rocprofv3 kernel-trace summary of one tools/kbench.py invocation:  bash tools/kprof.sh <tag> <kbench args...>
# (VIVIM_FWD_VARIANT / VIVIM_BWD_VARIANT are inherited by python3 directly: no env/bash hop behind rocprofv3)
set -e -o pipefail
tag=${1:?tag}; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kprof_$tag" -o k -- python3 "$root/tools/kbench.py" "$@" > "$out/kprof_$tag.log" 2>&1
cd "$root"
python tools/prof_summary.py "$out/kprof_$tag" "$out/kprof_${tag}_kernel_stats.csv" 24
rm -rf "$out/kprof_$tag"
cut -c1-110 "$out/kprof_${tag}_kernel_stats.csv" | cut -d, -f1-4 | head -16
