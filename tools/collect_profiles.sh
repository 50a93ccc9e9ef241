#!/bin/bash
# On the GPU box: everything profiles/rNN_* is made of, into gpurun_out/collect_<tag>/.  usage: tools/collect_profiles.sh r02
set -u
tag=${1:-run}
root=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun exports GRAFT_REPO_ROOT)}
out=$root/gpurun_out/collect_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root"
python3 bench.py > $out/bench.json 2> $out/bench.err
for c in 3 4 5; do python3 bench.py --config $c > $out/bench_config$c.json 2>> $out/bench.err; done
python3 bench.py --strong --no-cpu-baseline > $out/bench_strong_n1.json 2>> $out/bench.err
python3 bench.py --gpus 2 --rehearse-on-one-gpu --no-cpu-baseline > $out/bench_two_ranks_on_one_gpu.json 2>> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --no-cpu-baseline --no-other-configs > $out/bench_under_rocprof.json 2>&1
cp $out/prof/*/*_kernel_stats.csv $out/kernel_stats.csv
python3 tools/timed_region_average.py $out/prof/*/*_kernel_trace.csv 200 > $out/kernel_timed_region.json
for c in 3 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c$c -- python3 bench.py --no-cpu-baseline --config $c --steps 3 > $out/bench_config${c}_under_rocprof.json 2>&1
  cp $out/prof_c$c/*/*_kernel_stats.csv $out/kernel_stats_config$c.csv
done
python3 tools/measure_configs.py 2>/dev/null | tail -1 > $out/other_configs.json
# the catalogue driver end to end, with the share of instants at which its kernel stream was found empty
DLAGP_RUNS=3 DLAGP_POLL_IDLE=1 DLAGP_TIME_ACCUMULATE=1 python3 tools/measure_catalogue.py 2048 10000 32 2>/dev/null | tail -1 > $out/catalogue_pipeline.json
DLAGP_RUNS=3 DLAGP_POLL_IDLE=1 DLAGP_TIME_ACCUMULATE=1 python3 tools/measure_catalogue.py 2048 10000 32 dr12q 2>/dev/null | tail -1 > $out/catalogue_pipeline_dr12q.json
python3 tools/measure_catalogue.py 8192 10000 32 dr12q 2>/dev/null | tail -1 > $out/catalogue_pipeline_dr12q_8192.json
# one sightline of DR12Q's median length through bench.py's own step
python3 bench.py --pixels 685 --no-cpu-baseline --no-other-configs 2>> $out/bench.err | tail -1 > $out/bench_pixels_685.json
# efficiency against the sightline length (real DR12Q sightlines: n <= 1250, median ~685): one JSON per n_u, and the kernel
# stats of the same command under rocprofv3
echo '[' > $out/frac_vs_n.json
first=1
for n in 300 685 1250 3061; do
  [ $first = 1 ] || echo ',' >> $out/frac_vs_n.json
  first=0
  python3 tools/measure_frac_vs_n.py --pixels $n --stamps 2>> $out/bench.err | tail -1 >> $out/frac_vs_n.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_n$n -- python3 tools/measure_frac_vs_n.py --pixels $n > $out/frac_n${n}_under_rocprof.json 2>> $out/bench.err
  cp $out/prof_n$n/*/*_kernel_stats.csv $out/kernel_stats_n$n.csv
  rm -rf $out/prof_n$n
done
echo ']' >> $out/frac_vs_n.json
python3 tools/measure_frac_vs_n.py --pixels 755 --spread 0.65 2>> $out/bench.err | tail -1 > $out/frac_ragged_mix.json
# the drop-in route (the reference driver's loop, imports swapped): wall time (median of three), the same under rocprofv3 for
# its kernels, the kernels' timeline (chain span, device-busy time, host gap between spectra) and the host's functions
for i in 1 2 3; do python3 tools/measure_route_a.py 256 2>/dev/null | tail -1 > $out/route_a_wall_$i.json; done
python3 - "$out" <<'PY'
import json, sys
out = sys.argv[1]
runs = sorted((json.loads(open("%s/route_a_wall_%d.json" % (out, i)).read()) for i in (1, 2, 3)), key=lambda d: d["ms_per_spectrum"])
median = runs[1]
median["ms_per_spectrum_runs"] = [r["ms_per_spectrum"] for r in runs]
json.dump(median, open(out + "/route_a_wall.json", "w"))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_route_a -- python3 tools/measure_route_a.py 256 > $out/route_a_under_rocprof.json 2>> $out/bench.err
cp $out/prof_route_a/*/*_kernel_stats.csv $out/kernel_stats_route_a.csv
python3 tools/route_a_timeline.py $(ls $out/prof_route_a/*/*_kernel_trace.csv | head -1) > $out/route_a_timeline.json
python3 tools/route_a_frac.py $out/kernel_stats_route_a.csv $out/route_a_wall.json $out/route_a_timeline.json > $out/route_a.json
python3 tools/route_a_host_segments.py 256 > $out/route_a_host_segments.json 2>> $out/bench.err
# ... and the catalogue driver's kernels
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_cat -- python3 tools/measure_catalogue.py 1024 10000 32 dr12q > $out/catalogue_under_rocprof.json 2>> $out/bench.err
cp $out/prof_cat/*/*_kernel_stats.csv $out/kernel_stats_catalogue.csv
rm -rf $out/prof_route_a $out/prof_cat
rm -rf $out/prof $out/prof_c3 $out/prof_c4 $out/prof_c5
ls -la $out
