set -x
mkdir -p gpurun_out/r4
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4/gpu_tests.log 2>&1
tail -3 gpurun_out/r4/gpu_tests.log
python tools/measure_route_a.py 256 > gpurun_out/r4/route_a4.json 2> gpurun_out/r4/route_a4.err
cat gpurun_out/r4/route_a4.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/prof_route_a4 -o ra -- python3 tools/measure_route_a.py 64 > /dev/null 2>&1
cp $(find gpurun_out/r4/prof_route_a4 -name "*kernel_stats.csv" | head -1) gpurun_out/r4/route_a_kernel_stats4.csv
rm -rf gpurun_out/r4/prof_route_a4
timeout -k 10 300 python tools/soak_sweeps.py 13 100 > gpurun_out/r4/soak_sweeps2.log 2>&1
tail -1 gpurun_out/r4/soak_sweeps2.log
