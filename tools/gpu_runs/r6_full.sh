set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $out/gputest_full.log 2>&1; echo "pytest rc $?" >> $out/gputest_full.log
tail -22 $out/gputest_full.log
timeout -k 10 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 1500 $out/bench_default.json
