set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/collect_r06b; mkdir -p $out
for i in 1 2 3; do python3 tools/measure_route_a.py 256 2>/dev/null | tail -1 > $out/route_a_wall_$i.json; done
python3 - <<'PY'
import json
runs=[json.loads(open("gpurun_out/collect_r06b/route_a_wall_%d.json"%i).read()) for i in (1,2,3)]
runs.sort(key=lambda d:d["ms_per_spectrum"])
med=runs[1]; med["ms_per_spectrum_runs"]=[r["ms_per_spectrum"] for r in runs]
json.dump(med, open("gpurun_out/collect_r06b/route_a_wall.json","w"))
print(med["ms_per_spectrum_runs"])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_route_a -- python3 tools/measure_route_a.py 256 > $out/route_a_under_rocprof.json 2>/dev/null
cp $out/prof_route_a/*/*_kernel_stats.csv $out/kernel_stats_route_a.csv; rm -rf $out/prof_route_a
# where the wall time goes: the kernels' timeline (chain span, idle inside, host gap between spectra) and the host's functions
rocprofv3 --kernel-trace --output-format csv -d $out/prof_tl -- python3 tools/measure_route_a.py 256 > /dev/null 2>&1
python3 tools/route_a_timeline.py $(ls $out/prof_tl/*/*_kernel_trace.csv | head -1) > $out/route_a_timeline.json
rm -rf $out/prof_tl
python3 tools/route_a_frac.py $out/kernel_stats_route_a.csv $out/route_a_wall.json $out/route_a_timeline.json > $out/route_a.json
python3 tools/route_a_host_segments.py 256 > $out/route_a_host_segments.json 2>/dev/null
head -c 900 $out/route_a_timeline.json
