set -x
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --memory-copy-trace --stats -d gpurun_out/r4/prof_route_a -o ra -- python3 tools/measure_route_a.py 64 > gpurun_out/r4/route_a_traced.json 2> gpurun_out/r4/route_a_traced.err
ls gpurun_out/r4/prof_route_a/ > gpurun_out/r4/prof_ls.txt 2>&1
f=$(ls gpurun_out/r4/prof_route_a/*kernel_trace.csv | head -1)
python tools/trace_busy.py $f --window 0.14 > gpurun_out/r4/route_a_busy.txt 2>&1
find gpurun_out/r4/prof_route_a -name "*trace.csv" -size +20M -delete
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4/gpu_tests.log 2>&1
tail -5 gpurun_out/r4/gpu_tests.log
