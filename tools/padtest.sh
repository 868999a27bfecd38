set -x
python tools/kbench.py --config 2 --groups 3 --kernels sf,sb --iters 20
python tools/kbench.py --config 2 --groups 3 --kernels sf,sb --iters 20 --pad 64
VIVIM_FWD_VARIANT=6 VIVIM_BWD_VARIANT=4 python tools/kbench.py --config 2 --groups 3 --kernels sf,sb --stages 0 --iters 20
VIVIM_FWD_VARIANT=6 VIVIM_BWD_VARIANT=4 python tools/kbench.py --config 2 --groups 3 --kernels sf,sb --stages 0 --iters 20 --pad 64
VIVIM_FWD_VARIANT=6 VIVIM_BWD_VARIANT=4 python tools/kbench.py --config 3 --kernels sf,sb --stages 0 --iters 8
VIVIM_FWD_VARIANT=6 VIVIM_BWD_VARIANT=4 python tools/kbench.py --config 3 --kernels sf,sb --stages 0 --iters 8 --pad 32
python tools/kbench.py --config 3 --kernels sf,sb --stages 0 --iters 8
python tools/kbench.py --config 3 --kernels sf,sb --stages 0 --iters 8 --pad 32
