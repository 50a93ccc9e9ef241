set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for opts in "null_side=1" "null_side=0"; do
  DLAGP_OPTIONS=$opts timeout -k 10 300 python3 tools/measure_route_a.py 256 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('route A', d['options'], round(d['ms_per_spectrum'],4))"
done; done
