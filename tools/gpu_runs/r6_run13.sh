set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for opts in "pixel_splits=0" "pixel_splits=2" "pixel_splits=3" "pixel_splits=4" "plan_item_cost=5" "plan_item_cost=40" "plan_item_cost=80" "pixel_splits=0"; do
  for rep in 1 2; do DLAGP_OPTIONS=$opts timeout -k 10 300 python3 tools/measure_route_a.py 256 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('route A', d['options'], round(d['ms_per_spectrum'],4))"; done
done
