set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/ab_null_level; mkdir -p $out
for i in 1 2 3; do
  for lvl in -1 1 2 3; do
    DLAGP_OPTIONS="null_side_level=$lvl" python3 tools/measure_route_a.py 256 2>/dev/null | tail -1 > $out/level_${lvl}_$i.json
  done
  DLAGP_LIB=$PWD/gpy_dla_detection_amd/csrc/libdlagp_prev.so python3 tools/measure_route_a.py 256 2>/dev/null | tail -1 > $out/level_prev_$i.json
done
python3 - <<'PY'
import json
for tag in ("-1", "1", "2", "3", "prev"):
    print(tag, [round(json.load(open("gpurun_out/ab_null_level/level_%s_%d.json" % (tag, i)))["ms_per_spectrum"], 4) for i in (1, 2, 3)])
PY
