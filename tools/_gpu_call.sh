set -x
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cache or sweep or shared or cached or level or config" > gpurun_out/r4/gpu_tests_cache.log 2>&1
tail -3 gpurun_out/r4/gpu_tests_cache.log
timeout -k 10 600 python tools/soak_sweeps.py 12 150 > gpurun_out/r4/soak_sweeps.log 2>&1
tail -2 gpurun_out/r4/soak_sweeps.log
bash tools/ab_kernels.sh "--config 5 --steps 3" ab/base.so ab/new.so ab/new2.so > gpurun_out/r4/ab_config5.txt 2>&1
bash tools/ab_kernels.sh "--config 5 --steps 3" ab/base.so ab/new.so ab/new2.so >> gpurun_out/r4/ab_config5.txt 2>&1
cat gpurun_out/r4/ab_config5.txt
bash tools/ab_catalogue_kernels.sh "512" ab/base.so ab/new.so ab/new2.so > gpurun_out/r4/ab_catalogue.txt 2>&1
cat gpurun_out/r4/ab_catalogue.txt
