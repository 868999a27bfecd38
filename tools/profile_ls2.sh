#!/bin/bash
# Round-3 evidence for the second-generation lanes=states backward: stamps, SQ counters, HBM counters, clock.
#   gpurun --timeout 900 -- 'bash tools/profile_ls2.sh'      -> gpurun_out/r03_*; copy what is to be kept to profiles/
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd "$root"
export VIVIM_FWD_VARIANT=5 VIVIM_BWD_VARIANT=5
timeout -k 5 60 tools/ls2_lab 3 768 5120 3 > "$out/r03_ls2_stamps.log" 2>&1
bash tools/kpmc.sh r03_ls2_cfg2s1 --config 2 --groups 3 --kernels sb --stages 1 > /dev/null 2>&1; cp "$out/kpmc_r03_ls2_cfg2s1.txt" "$out/r03_pmc_sq_ls2_cfg2_stage1.txt"
bash tools/kpmc2.sh r03_ls2_cfg2s1 --config 2 --groups 3 --kernels sb --stages 1 > /dev/null 2>&1; cat "$out/kpmc2_r03_ls2_cfg2s1.txt" >> "$out/r03_pmc_sq_ls2_cfg2_stage1.txt"
bash tools/khbm.sh r03_ls2_cfg3s1 --config 3 --groups 3 --kernels sb --stages 1 --iters 3 > /dev/null 2>&1; cp "$out/khbm_r03_ls2_cfg3s1.txt" "$out/r03_hbm_counters_ls2_cfg3_stage1.txt"
bash tools/khbm.sh r03_ls2_cfg2s0 --config 2 --groups 3 --kernels sb --stages 0,1 --iters 3 > /dev/null 2>&1; cp "$out/khbm_r03_ls2_cfg2s0.txt" "$out/r03_hbm_counters_ls2_cfg2_stages01.txt"
bash tools/kclk.sh r03_ls2 --config 3 --groups 3 --kernels sf,sb --stages 0 --iters 3 > "$out/r03_clock_cfg3_grouped_stage0.txt" 2>&1
bash tools/kprof.sh r03_ls2_cfg2 --config 2 --groups 3 --kernels sb > /dev/null 2>&1
ls -la "$out" | grep r03_
