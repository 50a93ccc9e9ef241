set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/host_segments; mkdir -p $out
python3 tools/route_a_host_segments.py 256 > $out/segments.json 2> $out/segments.err
rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $out/prof -- python3 tools/measure_route_a.py 64 > $out/under_rocprof.json 2>/dev/null
ls $out/prof/*/ > $out/files.txt
api=$(ls $out/prof/*/*_hip_api_trace.csv | head -1)
python3 - "$api" > $out/one_spectrum_api.txt <<'PY'
import csv, sys, collections
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r["Thread_Id"]) for r in csv.DictReader(open(sys.argv[1]))))
# spectra are separated by the wait for the results: hipEventSynchronize / hipStreamSynchronize of > 200 us
waits = [i for i, r in enumerate(rows) if r[2] in ("hipEventSynchronize", "hipStreamSynchronize", "hipDeviceSynchronize") and r[1] - r[0] > 200000]
print("long waits", len(waits))
a, b = waits[40], waits[41]
t0 = rows[a][1]
prev = t0
for s, e, fn, th in rows[a + 1:b + 1]:
    print("%8.1f  +%6.1f  %6.1f us  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, fn))
    prev = e
# totals per function per spectrum over 20 spectra
tot = collections.Counter(); cnt = collections.Counter()
for s, e, fn, th in rows[waits[20] + 1:waits[40] + 1]:
    tot[fn] += e - s; cnt[fn] += 1
print()
for fn, t in tot.most_common():
    print("%-34s %6.1f calls  %8.1f us per spectrum" % (fn, cnt[fn] / 20, t / 20 / 1e3))
PY
rm -rf $out/prof
head -c 3000 $out/segments.json
