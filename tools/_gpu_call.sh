set -x
mkdir -p gpurun_out/r4
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4/gpu_tests.log 2>&1
tail -3 gpurun_out/r4/gpu_tests.log
python tools/measure_route_a.py 256 > gpurun_out/r4/route_a3.json 2> gpurun_out/r4/route_a3.err
cat gpurun_out/r4/route_a3.json
bash tools/collect_pmc.sh r04 > gpurun_out/r4/pmc.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in base new; do
  for group in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    DLAGP_LIB=$PWD/ab/$lib.so bash tools/pmc_pass.sh gpurun_out/r4/pmc5_${lib} "$group" -- python3 bench.py --no-cpu-baseline --config 5 --steps 2 --warmup 1 --ramp-ms 0 >> gpurun_out/r4/pmc5_${lib}.txt 2>> gpurun_out/r4/pmc5_err.txt
    rm -rf gpurun_out/r4/pmc5_${lib}
  done
done
grep -A4 "false, 2, 0" gpurun_out/r4/pmc5_base.txt gpurun_out/r4/pmc5_new.txt | head -60
