set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "level_finish or resampling or by_sample or device_sweep or config5 or settles or golden or class_api or reference_driver" > $out/t12.log 2>&1; echo "pytest rc $?" >> $out/t12.log
tail -4 $out/t12.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_g -- python3 tools/measure_route_a.py 128 > $out/g.json 2>/dev/null
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r6/prof_g/*/*_kernel_stats.csv")[0]
rows={r["Name"].split("(")[0]:r for r in csv.DictReader(open(f))}
print({k.replace("dlagp::",""):round(float(v["AverageNs"])/1e3,1) for k,v in rows.items() if "level_" in k})
PY
rm -rf $out/prof_g
for i in 1 2 3; do timeout -k 10 300 python3 tools/measure_route_a.py 256 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('route A', round(d['ms_per_spectrum'],4))"; done
