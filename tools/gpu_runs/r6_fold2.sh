set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/fold; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "finalisation_kernels_tail" > $out/pytest2.log 2>&1
echo "pytest rc $?" >> $out/pytest2.log
tail -5 $out/pytest2.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --no-cpu-baseline --no-other-configs > $out/under_rocprof.json 2>&1
cp $out/prof/*/*_kernel_stats.csv $out/kernel_stats_folded.csv; rm -rf $out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --option fold_evidence=0 --no-cpu-baseline --no-other-configs > $out/under_rocprof0.json 2>&1
cp $out/prof/*/*_kernel_stats.csv $out/kernel_stats_apart.csv; rm -rf $out/prof
cut -d, -f1-4 $out/kernel_stats_folded.csv | cut -c1-120 | head -6
cut -d, -f1-4 $out/kernel_stats_apart.csv | cut -c1-120 | head -6
