set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/ab_prefix; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "model_evidences or driver_loop or model_selection or pair or one_call or batch or null" > $out/pytest.log 2>&1
echo "pytest rc $?" >> $out/pytest.log
tail -3 $out/pytest.log
for i in 1 2 3; do
  DLAGP_LIB=$PWD/gpy_dla_detection_amd/csrc/libdlagp_prev.so python3 tools/measure_route_a.py 256 2>/dev/null | tail -1 > $out/prev_$i.json
  python3 tools/measure_route_a.py 256 2>/dev/null | tail -1 > $out/new_$i.json
done
python3 - <<'PY'
import json
for tag in ("prev", "new"):
    print(tag, [round(json.load(open("gpurun_out/ab_prefix/%s_%d.json" % (tag, i)))["ms_per_spectrum"], 4) for i in (1, 2, 3)])
PY
