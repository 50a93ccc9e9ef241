set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/batchsize; mkdir -p $out
for bs in 32 48 64 96; do
  DLAGP_RUNS=3 python3 tools/measure_catalogue.py 4096 10000 $bs dr12q 2>/dev/null | tail -1 > $out/bs_$bs.json
done
python3 - <<'PY'
import json
for bs in (32, 48, 64, 96):
    d = json.load(open("gpurun_out/batchsize/bs_%d.json" % bs))
    print(bs, d.get("spectra_per_s"), d.get("spectra_per_s_runs"), d.get("whole_run_frac_of_fp64_peak"))
PY
