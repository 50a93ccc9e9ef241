set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "native_batch or process_qso or pipelined or reference_driver" > $out/t2.log 2>&1; echo "pytest rc $?" >> $out/t2.log
tail -15 $out/t2.log
for native in 1 0; do
  DLAGP_NATIVE_PREPARE=$native DLAGP_RUNS=3 DLAGP_POLL_IDLE=1 timeout -k 10 600 python3 tools/measure_catalogue.py 2048 10000 32 dr12q 2>$out/cat_native$native.err | tail -1 > $out/cat_native$native.json
  python3 -c "
import json; d=json.load(open('$out/cat_native$native.json')); print('native', $native, {k: d[k] for k in d if k in ('spectra_per_s_runs','spectra_per_s_median','whole_run_frac_of_fp64_peak','kernel_stream_idle')})"
done
