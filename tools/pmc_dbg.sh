#!/bin/bash
# one WRITE_SIZE pass with the Python fault handler on, to see where a host-side fault under the profiler comes from
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_dbg" -o b -- python3 -X faulthandler "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-by-config > "$out/pmc_dbg.log" 2>&1
echo "rc $?"
grep -n -A25 "Fatal Python error\|Current thread" "$out/pmc_dbg.log" | head -60
rm -rf "$out/pmc_dbg"
