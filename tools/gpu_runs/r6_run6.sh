set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6; mkdir -p $out
tag=${1:-x}
timeout -k 10 300 python3 tools/measure_route_a.py 256 2>/dev/null | tail -1 > $out/route_a_$tag.json; cat $out/route_a_$tag.json
DLAGP_RUNS=3 DLAGP_POLL_IDLE=1 DLAGP_TIME_ACCUMULATE=1 timeout -k 10 600 python3 tools/measure_catalogue.py 2048 10000 32 dr12q 2>$out/cat_$tag.err | tail -1 > $out/cat_$tag.json
python3 - $tag <<'PY'
import json,sys
tag=sys.argv[1]
d=json.load(open("gpurun_out/r6/cat_%s.json"%tag))
print(tag, {k:d[k] for k in d if k in ("spectra_per_s_runs","spectra_per_s_median","accumulate_share_of_run","accumulate_frac_of_fp64_peak","whole_run_frac_of_fp64_peak","kernel_stream_idle")})
PY
