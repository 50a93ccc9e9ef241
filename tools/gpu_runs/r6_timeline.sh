set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/timeline; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/prof -- python3 tools/measure_route_a.py 256 > $out/under_rocprof.json 2>/dev/null
trace=$(ls $out/prof/*/*_kernel_trace.csv | head -1)
head -1 $trace > $out/trace_header.txt
python3 tools/route_a_timeline.py $trace > $out/timeline.json
python3 tools/route_a_timeline.py $trace 60 > $out/timeline_60.json
# one chain in full, for reading: the 100th spectrum's kernels with start offsets
python3 - "$trace" > $out/one_chain.txt <<'PY'
import csv, sys, re
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", ""), r.get("Queue_Id", "")) for r in csv.DictReader(open(sys.argv[1]))))
chains, cur, edge = [], [], None
for row in rows:
    if cur and row[0] - edge > 120000:
        chains.append(cur); cur = []
    edge = row[1] if not cur else max(edge, row[1])
    cur.append(row)
chains.append(cur)
c = chains[100]
t0 = c[0][0]
prev_end = t0
for s, e, name, stream, queue in c:
    print("%8.1f %7.1f gap %6.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, queue, re.sub(r"\(.*", "", name)[:110]))
    prev_end = max(prev_end, e)
print("next chain starts", (chains[101][0][0] - t0) / 1e3)
PY
rm -rf $out/prof
cat $out/timeline.json
