set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6; mkdir -p $out
tag=${1:-packed}
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $out/gputest_$tag.log 2>&1; echo "pytest rc $?" >> $out/gputest_$tag.log
tail -4 $out/gputest_$tag.log
timeout -k 10 300 python3 tools/measure_route_a.py 256 2>/dev/null | tail -1 > $out/route_a_$tag.json; cat $out/route_a_$tag.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_a -- python3 tools/measure_route_a.py 256 > $out/route_a_prof_$tag.json 2>$out/route_a_prof_$tag.err
cp $out/prof_a/*/*_kernel_stats.csv $out/route_a_kernel_stats_$tag.csv; rm -rf $out/prof_a
DLAGP_RUNS=3 DLAGP_POLL_IDLE=1 DLAGP_TIME_ACCUMULATE=1 timeout -k 10 600 python3 tools/measure_catalogue.py 2048 10000 32 dr12q 2>$out/cat_$tag.err | tail -1 > $out/cat_$tag.json
python3 - $tag <<'PY'
import json,sys,csv
tag=sys.argv[1]
d=json.load(open("gpurun_out/r6/cat_%s.json"%tag))
print(tag, {k:d[k] for k in d if k in ("spectra_per_s_runs","spectra_per_s_median","accumulate_share_of_run","accumulate_frac_of_fp64_peak","whole_run_frac_of_fp64_peak","kernel_stream_idle")})
rows=list(csv.DictReader(open("gpurun_out/r6/route_a_kernel_stats_%s.csv"%tag)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("route A kernel ms per spectrum", tot/1e6/258)
for r in rows[:12]:
    print("  %-70s %6s %9.1f us avg  %5.1f%%"%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
