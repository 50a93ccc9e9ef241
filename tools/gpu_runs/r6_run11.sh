set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for opts in "setup_elements_per_lane=1" "setup_elements_per_lane=4" "setup_elements_per_lane=2" "setup_elements_per_lane=1" "setup_elements_per_lane=4"; do
  DLAGP_OPTIONS=$opts DLAGP_RUNS=3 DLAGP_POLL_IDLE=1 DLAGP_TIME_ACCUMULATE=1 timeout -k 10 600 python3 tools/measure_catalogue.py 2048 10000 32 dr12q 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('catalogue', d['options'], [round(x) for x in d['spectra_per_s_runs']], 'idle', round(d['kernel_stream_idle']['share'],3), 'acc frac', round(d['accumulate_frac_of_fp64_peak'],3), 'share', round(d['accumulate_share_of_run'],3))"
done
