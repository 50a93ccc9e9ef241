set -x
mkdir -p gpurun_out/r4
python tools/measure_route_a.py 64 > gpurun_out/r4/route_a2.json 2> gpurun_out/r4/route_a2.err
python tools/measure_route_a.py 256 > gpurun_out/r4/route_a2_256.json 2>> gpurun_out/r4/route_a2.err
DLAGP_PROFILE=1 python tools/measure_route_a.py 64 > gpurun_out/r4/route_a2_prof.log 2>&1
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4/gpu_tests.log 2>&1
tail -5 gpurun_out/r4/gpu_tests.log
