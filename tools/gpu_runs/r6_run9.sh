set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6; mkdir -p $out
for g in 16 8 32 4 16; do
  DLAGP_OPTIONS=level_search_groups=$g timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_g -- python3 tools/measure_route_a.py 128 > $out/g.json 2>/dev/null
  python3 - $g <<'PY'
import csv,glob,sys
f=glob.glob("gpurun_out/r6/prof_g/*/*_kernel_stats.csv")[0]
rows={r["Name"].split("(")[0]:r for r in csv.DictReader(open(f))}
print("groups", sys.argv[1], {k.replace("dlagp::",""):round(float(v["AverageNs"])/1e3,1) for k,v in rows.items() if "level_" in k})
PY
  rm -rf $out/prof_g
done
