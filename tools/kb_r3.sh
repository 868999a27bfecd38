# Round-3 comparison of the backward families at the grouped shapes (tools/kbench.py timings).
# bwd 4 = first-generation lanes=states, 5 = second generation (scan_ls2.hip), 2 = lanes=tokens (round 1)
for cfg in 2 3; do
  it=20; st=0,1,2,3; [ $cfg = 3 ] && it=5 && st=0,1
  echo "== cfg $cfg: fwd 5 (16-token checkpoints everywhere), bwd 5 (ls2)"
  VIVIM_FWD_VARIANT=5 VIVIM_BWD_VARIANT=5 python tools/kbench.py --config $cfg --groups 3 --kernels sb --stages $st --iters $it
  echo "== cfg $cfg: fwd 5, bwd 4 (ls1)"
  VIVIM_FWD_VARIANT=5 VIVIM_BWD_VARIANT=4 python tools/kbench.py --config $cfg --groups 3 --kernels sb --stages $st --iters $it
  echo "== cfg $cfg: automatic"
  python tools/kbench.py --config $cfg --groups 3 --kernels sb --stages $st --iters $it
done 2>&1 | grep -v amdgpu.ids
