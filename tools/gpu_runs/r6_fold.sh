set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/fold; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "finalisation_kernels_tail or graph_capturable" > $out/pytest.log 2>&1
echo "pytest rc $?" >> $out/pytest.log
tail -5 $out/pytest.log
for i in 1 2 3; do
  python3 bench.py --option fold_evidence=0 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $out/apart_$i.json
  python3 bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $out/folded_$i.json
done
python3 - <<'PY'
import json
for tag in ("apart", "folded"):
    rows = [json.load(open("gpurun_out/fold/%s_%d.json" % (tag, i))) for i in (1, 2, 3)]
    print(tag, [round(r["ms_per_step"], 4) for r in rows], [round(r["roofline"]["kernel_ms"], 4) for r in rows], rows[0].get("config"))
PY
