set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6; mkdir -p $out
for seed in 1 2 3; do timeout -k 10 400 python3 tools/soak_levels.py $seed 150 2>/dev/null | tail -1; done
timeout -k 10 500 python3 tools/soak_pipeline.py 1024 10000 3 2>/dev/null | tail -1 | cut -c1-900
timeout -k 10 300 python3 tools/soak_sweeps.py 5 60 2>/dev/null | tail -2
