set -x
mkdir -p gpurun_out/r4
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4/gpu_tests.log 2>&1
tail -3 gpurun_out/r4/gpu_tests.log
