set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for opts in "pack_setup_uploads=2" "pack_setup_uploads=0" "pack_setup_uploads=2" "pack_setup_uploads=0"; do
  for rep in 1 2; do DLAGP_OPTIONS=$opts timeout -k 10 300 python3 tools/measure_route_a.py 256 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('route A', d['options'], round(d['ms_per_spectrum'],4))"; done
  DLAGP_OPTIONS=$opts DLAGP_RUNS=3 DLAGP_POLL_IDLE=1 timeout -k 10 600 python3 tools/measure_catalogue.py 2048 10000 32 dr12q 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('catalogue', d['options'], [round(x) for x in d['spectra_per_s_runs']], round(d['kernel_stream_idle']['share'],3))"
done
