set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r06 > gpurun_out/collect_r06.log 2>&1
tail -5 gpurun_out/collect_r06.log
cat gpurun_out/collect_r06/bench.json | cut -c1-600
