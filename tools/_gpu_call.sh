set -x
mkdir -p gpurun_out/r4
cat /sys/fs/cgroup/cpu.max > gpurun_out/r4/cpuinfo.txt 2>&1
nproc >> gpurun_out/r4/cpuinfo.txt; uptime >> gpurun_out/r4/cpuinfo.txt
python bench.py --cpu-baseline-only > gpurun_out/r4/cpu_a.json 2> gpurun_out/r4/cpu_a.err
python bench.py --cpu-baseline-only > gpurun_out/r4/cpu_b.json 2> gpurun_out/r4/cpu_b.err
python bench.py > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err && \
timeout -k 10 1000 python -m pytest tests/test_gpu_distributed.py -x -q -m gpu > gpurun_out/r4/dist_tests.log 2>&1
tail -5 gpurun_out/r4/dist_tests.log
