set -x
mkdir -p gpurun_out/r4
bash tools/collect_profiles.sh r04 > gpurun_out/r4/collect.log 2>&1
tail -3 gpurun_out/r4/collect.log
