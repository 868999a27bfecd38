#!/bin/bash
# Build timing-ablation variants of the lanes=states backward (scan_ls.hip, -DLS_ABL=n: results are wrong, only the time
# means something) next to the product library:  bash tools/abl.sh build 1 2 3 ...   /   bash tools/abl.sh run 1 2 3 ...
set -e
cd "$(dirname "$0")/../vivim_amd/csrc"
mode=$1; shift
mkdir -p abl
if [ "$mode" = chanbuild ]; then     # lanes=channels forward ablations (-DCH_ABL=n)
  for n in "$@"; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DCH_ABL=$n -c scan_fwd_chan.hip -o abl/scan_fwd_chan_$n.o 2> abl/scan_fwd_chan_$n.res
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/libvivim_chabl_$n.so capi.o conv1d.o conv1d_cl.o scan_fwd.o abl/scan_fwd_chan_$n.o scan_ls.o scan_ls2.o scan_bwd.o dwconv.o dirmap.o update.o
    echo "built chan abl $n"
  done
  exit 0
fi
if [ "$mode" = bwdbuild ]; then      # round-1 backward without its dB / dC atomics (-DBW_ABL=1)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DBW_ABL=1 -c scan_bwd.hip -o abl/scan_bwd_1.o 2> abl/scan_bwd_1.res
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/libvivim_bwabl_1.so capi.o conv1d.o conv1d_cl.o scan_fwd.o scan_fwd_chan.o scan_ls.o scan_ls2.o abl/scan_bwd_1.o dwconv.o dirmap.o update.o
  echo "built bwd abl 1"; exit 0
fi
if [ "$mode" = bwdrun ]; then
  cd ../..
  for lib in libvivim_hip.so abl/libvivim_bwabl_1.so; do
    echo "== $lib"
    VIVIM_LIB=$PWD/vivim_amd/csrc/$lib python tools/kbench.py --config 3 --stages 0 --kernels sb --iters 6 | grep stage
    VIVIM_LIB=$PWD/vivim_amd/csrc/$lib python tools/kbench.py --config 2 --groups 3 --stages 0 --kernels sb --iters 20 | grep stage
  done
  exit 0
fi
if [ "$mode" = chanrun ]; then
  cd ../..
  for n in "$@"; do
    echo "== CH_ABL=$n"
    VIVIM_LIB=$PWD/vivim_amd/csrc/abl/libvivim_chabl_$n.so python tools/kbench.py --config 3 --stages 0 --kernels sf --iters 6 | grep stage
    VIVIM_LIB=$PWD/vivim_amd/csrc/abl/libvivim_chabl_$n.so python tools/kbench.py --config 2 --groups 3 --stages 0 --kernels sf --iters 20 | grep stage
  done
  exit 0
fi
if [ "$mode" = dirbuild ]; then      # direction maps without the identity / flip copies (-DDIR_ABL=1)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DDIR_ABL=1 -c dirmap.hip -o abl/dirmap_1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/libvivim_dirabl_1.so capi.o conv1d.o conv1d_cl.o scan_fwd.o scan_fwd_chan.o scan_ls.o scan_ls2.o scan_bwd.o dwconv.o abl/dirmap_1.o update.o
  echo "built dir abl 1"; exit 0
fi
if [ "$mode" = ls2build ]; then      # second-generation lanes=states backward (-DLS2_ABL=n)
  for n in "$@"; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DLS2_ABL=$n -Rpass-analysis=kernel-resource-usage -c scan_ls2.hip -o abl/scan_ls2_$n.o 2> abl/scan_ls2_$n.res
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/libvivim_ls2abl_$n.so capi.o conv1d.o conv1d_cl.o scan_fwd.o scan_fwd_chan.o scan_ls.o abl/scan_ls2_$n.o scan_bwd.o dwconv.o dirmap.o update.o
    echo "built ls2 abl $n"
  done
  exit 0
fi
if [ "$mode" = ls2run ]; then
  cd ../..
  for n in "$@"; do
    echo "== LS2_ABL=$n"
    lib=$PWD/vivim_amd/csrc/abl/libvivim_ls2abl_$n.so; [ $n = 0 ] && lib=$PWD/vivim_amd/csrc/libvivim_hip.so
    VIVIM_LIB=$lib VIVIM_FWD_VARIANT=5 VIVIM_BWD_VARIANT=5 python tools/kbench.py --config 3 --groups 3 --stages 1 --kernels sb --iters 6 | grep stage
    VIVIM_LIB=$lib VIVIM_FWD_VARIANT=5 VIVIM_BWD_VARIANT=5 python tools/kbench.py --config 2 --groups 3 --stages 0,1 --kernels sb --iters 20 | grep stage
  done
  exit 0
fi
if [ "$mode" = build ]; then
  for n in "$@"; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DLS_ABL=$n -Rpass-analysis=kernel-resource-usage -c scan_ls.hip -o abl/scan_ls_$n.o 2> abl/scan_ls_$n.res
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/libvivim_abl_$n.so capi.o conv1d.o conv1d_cl.o scan_fwd.o scan_fwd_chan.o abl/scan_ls_$n.o scan_ls2.o scan_bwd.o dwconv.o dirmap.o update.o
    echo "built abl $n"
  done
else
  cd ../..
  for n in "$@"; do
    echo "== LS_ABL=$n"
    VIVIM_LIB=$PWD/vivim_amd/csrc/abl/libvivim_abl_$n.so VIVIM_BWD_VARIANT=4 python tools/kbench.py --config 3 --stages 0 --kernels sb --iters 6 | grep stage
    VIVIM_LIB=$PWD/vivim_amd/csrc/abl/libvivim_abl_$n.so VIVIM_BWD_VARIANT=4 python tools/kbench.py --config 2 --groups 3 --stages 1 --kernels sb --iters 20 | grep stage
  done
fi
