set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6; mkdir -p $out
: > $out/soak_levels_long.json
for seed in 21 22 23 24 25 26; do timeout -k 10 900 python3 tools/soak_levels.py $seed 400 2>/dev/null | tail -1 >> $out/soak_levels_long.json; tail -1 $out/soak_levels_long.json; done
timeout -k 10 600 python3 tools/soak_sweeps.py 9 200 2>/dev/null | tail -1 > $out/soak_sweeps.txt; cat $out/soak_sweeps.txt
timeout -k 10 600 python3 tools/soak_paths.py 9 300 2>/dev/null | tail -1 > $out/soak_paths.txt; cat $out/soak_paths.txt
