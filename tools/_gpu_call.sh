set -x
mkdir -p gpurun_out/r4
timeout -k 10 900 python tools/soak_paths.py 41 400 > gpurun_out/r4/soak_paths.log 2>&1
tail -2 gpurun_out/r4/soak_paths.log
timeout -k 10 600 python tools/soak_sweeps.py 42 300 > gpurun_out/r4/soak_sweeps3.log 2>&1
tail -1 gpurun_out/r4/soak_sweeps3.log
