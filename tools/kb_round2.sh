set -x
python tools/kbench.py --config 2 --groups 3 --kernels sf,sb
VIVIM_FWD_VARIANT=6 python tools/kbench.py --config 2 --groups 3 --kernels sf
VIVIM_FWD_VARIANT=1 VIVIM_BWD_VARIANT=2 python tools/kbench.py --config 2 --groups 3 --kernels sf,sb --stages 0
python tools/kbench.py --config 3 --kernels sf,sb --stages 0,1
VIVIM_FWD_VARIANT=6 python tools/kbench.py --config 3 --kernels sf --stages 0,1
python tools/kbench.py --config 3 --groups 3 --kernels sf,sb --stages 0 --iters 5
python tools/kbench.py --config 5 --kernels sf,sb
