set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $out/t18.log 2>&1; echo "pytest rc $?" >> $out/t18.log
tail -3 $out/t18.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_g -- python3 tools/measure_route_a.py 128 > $out/g.json 2>/dev/null
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r6/prof_g/*/*_kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel ms per spectrum", tot/1e6/130, "launches per spectrum", sum(int(r["Calls"]) for r in rows)/130)
print([(r["Name"][7:30], r["Calls"], round(float(r["AverageNs"])/1e3,1)) for r in rows if "level_" in r["Name"]])
PY
rm -rf $out/prof_g
for i in 1 2 3; do timeout -k 10 300 python3 tools/measure_route_a.py 256 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('route A', round(d['ms_per_spectrum'],4))"; done
