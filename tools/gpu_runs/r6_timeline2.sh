set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/timeline2; mkdir -p $out
rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $out/prof -- python3 tools/measure_route_a.py 128 > $out/under_rocprof.json 2>/dev/null
trace=$(ls $out/prof/*/*_kernel_trace.csv | head -1)
api=$(ls $out/prof/*/*_hip_api_trace.csv | head -1)
python3 tools/route_a_timeline.py $trace > $out/timeline.json
python3 - "$trace" "$api" > $out/one_spectrum.txt <<'PY'
import csv, sys, re
k = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K  q%s %s" % (r.get("Queue_Id", ""), re.sub(r"\(.*", "", r["Kernel_Name"])[:90])) for r in csv.DictReader(open(sys.argv[1]))]
a = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "  api " + r["Function"]) for r in csv.DictReader(open(sys.argv[2]))
     if r["Function"] not in ("hipGetDevice", "hipGetLastError", "hipSetDevice", "hipGetDeviceCount")]
rows = sorted(k + a)
waits = [i for i, r in enumerate(rows) if "Synchronize" in r[2] and r[1] - r[0] > 200000]
lo, hi = waits[60], waits[61]
t0 = rows[lo][1]
for s, e, name in rows[lo + 1:hi + 1]:
    print("%8.1f %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, name))
PY
rm -rf $out/prof
cat $out/timeline.json | head -40
