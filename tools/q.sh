set -x
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "families or full_size or strided or golden" 2>&1 | tail -2
python tools/kbench.py --config 2 --groups 3 --kernels sf --iters 30 | grep stage
python tools/kbench.py --config 3 --kernels sf --stages 0,1 --iters 8 | grep stage
