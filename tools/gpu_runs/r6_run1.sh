set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "level_finish or resampling or by_sample or device_sweep or config or full_size or settles or golden or class_api" > $out/t1.log 2>&1; echo "pytest rc $?" >> $out/t1.log
tail -25 $out/t1.log
for spread in 1 0; do
  DLAGP_OPTIONS=level_spread=$spread timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_a$spread -- python3 tools/measure_route_a.py 128 > $out/route_a_spread$spread.json 2>$out/route_a_spread$spread.err
  cp $out/prof_a$spread/*/*_kernel_stats.csv $out/route_a_kernel_stats_spread$spread.csv; rm -rf $out/prof_a$spread
  tail -1 $out/route_a_spread$spread.json
done
