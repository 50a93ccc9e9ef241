#!/bin/bash
# On the GPU box: GPU tests, then the bench under rocprofv3 kernel stats. Usage: tools/gpu_check.sh <tag>
tag=${1:-run}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu.log 2>&1
tail -3 gpurun_out/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_${tag}_bench.log 2>&1
grep "^{" gpurun_out/prof_${tag}_bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('evals/s %.3e  ms/step %.3f  kernel_ms %.3f  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
python3 - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob('gpurun_out/prof_%s/*/*_kernel_stats.csv' % sys.argv[1])[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print(r['Name'][:58].ljust(58), r['Calls'], "%.1f us avg" % (float(r['AverageNs']) / 1e3), r['Percentage'])
PY
