set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/collect_r06
bash tools/collect_pmc.sh r06 > gpurun_out/collect_pmc_r06.log 2>&1
tail -3 gpurun_out/collect_pmc_r06.log
for seed in 11 12; do python3 tools/soak_levels.py $seed 150 2>/dev/null | tail -1; done > gpurun_out/collect_r06/soak_levels.json
python3 tools/soak_pipeline.py 1024 10000 3 2>/dev/null | tail -1 > gpurun_out/collect_r06/soak_pipeline.json
python -m pytest tests -q -m gpu 2>&1 | tail -16 > gpurun_out/collect_r06/gputest_tail.txt
tail -3 gpurun_out/collect_r06/gputest_tail.txt
