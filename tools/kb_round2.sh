# Round-2 kernel comparison: lanes=states family (fwd 6 / bwd 4) against the round-1 families, tools/kbench.py timings.
set -x
VIVIM_FWD_VARIANT=6 VIVIM_BWD_VARIANT=4 python tools/kbench.py --config 2 --groups 3 --kernels sf,sb
VIVIM_FWD_VARIANT=5 VIVIM_BWD_VARIANT=4 python tools/kbench.py --config 2 --groups 3 --kernels sf
VIVIM_FWD_VARIANT=1 VIVIM_BWD_VARIANT=2 python tools/kbench.py --config 2 --groups 3 --kernels sf,sb
VIVIM_FWD_VARIANT=6 VIVIM_BWD_VARIANT=4 python tools/kbench.py --config 3 --kernels sf,sb --iters 10
VIVIM_FWD_VARIANT=5 python tools/kbench.py --config 3 --kernels sf --iters 10
VIVIM_FWD_VARIANT=1 VIVIM_BWD_VARIANT=2 python tools/kbench.py --config 3 --kernels sf,sb --iters 10
VIVIM_FWD_VARIANT=6 VIVIM_BWD_VARIANT=4 python tools/kbench.py --config 3 --groups 3 --kernels sf,sb --stages 0 --iters 5
VIVIM_FWD_VARIANT=1 VIVIM_BWD_VARIANT=2 python tools/kbench.py --config 3 --groups 3 --kernels sf,sb --stages 0 --iters 5
VIVIM_FWD_VARIANT=6 VIVIM_BWD_VARIANT=4 python tools/kbench.py --config 5 --kernels sf,sb
VIVIM_FWD_VARIANT=1 VIVIM_BWD_VARIANT=2 python tools/kbench.py --config 5 --kernels sf,sb
