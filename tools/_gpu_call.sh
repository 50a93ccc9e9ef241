set -x
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ranks_above_20" > gpurun_out/r4/gpu_tests_k.log 2>&1
tail -15 gpurun_out/r4/gpu_tests_k.log
