set -x
mkdir -p gpurun_out/r4
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4/gpu_tests.log 2>&1
tail -2 gpurun_out/r4/gpu_tests.log
python tools/measure_route_a.py 256 2>/dev/null | tail -1 > gpurun_out/r4/route_a6.json
cat gpurun_out/r4/route_a6.json
