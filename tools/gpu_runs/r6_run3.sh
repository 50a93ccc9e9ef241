set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6; mkdir -p $out
DLAGP_RUNS=2 DLAGP_POLL_IDLE=1 DLAGP_TIME_ACCUMULATE=1 timeout -k 10 600 python3 tools/measure_catalogue.py 2048 10000 32 dr12q 2>$out/cat_acc.err | tail -1 > $out/cat_acc.json
timeout -k 10 600 python3 tools/measure_catalogue.py 2048 10000 32 dr12q sweeps-only 2>>$out/cat_acc.err | tail -1 > $out/cat_sweeps_only.json
python3 - <<'PY'
import json
for f in ("cat_acc","cat_sweeps_only"):
    d=json.load(open("gpurun_out/r6/%s.json"%f))
    print(f, {k:d[k] for k in d if k in ("spectra_per_s","spectra_per_s_runs","accumulate_share_of_run","accumulate_frac_of_fp64_peak","whole_run_frac_of_fp64_peak","kernel_stream_idle","sweeps_only","kept_pixels")})
PY
