set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/soak_final; mkdir -p $out
timeout -k 10 600 python3 tools/soak_sweeps.py 21 200 2>/dev/null | tail -1 > $out/soak_sweeps.txt; cat $out/soak_sweeps.txt
timeout -k 10 600 python3 tools/soak_paths.py 21 300 2>/dev/null | tail -1 > $out/soak_paths.txt; cat $out/soak_paths.txt
timeout -k 10 600 python3 tools/soak_pipeline.py 1024 10000 3 2>/dev/null | tail -1 | cut -c1-300
for seed in 31 32 33 34; do timeout -k 10 300 python3 tools/soak_levels.py $seed 200 2>/dev/null | tail -1; done > $out/soak_levels.json; cat $out/soak_levels.json
