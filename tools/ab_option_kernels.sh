#!/bin/bash
# On the GPU box: per-kernel averages (rocprofv3 kernel stats) of one bench configuration for several values of a library option.
# usage: tools/ab_option_kernels.sh "<bench args>" <name> <v0> <v1> ...
ARGS=$1; NAME=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for v in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abo_$v -- python3 bench.py --no-cpu-baseline $ARGS --option $NAME=$v > gpurun_out/abo_$v.log 2>&1
  python3 - $v <<'PY'
import csv, glob, sys
f = glob.glob('gpurun_out/abo_%s/*/*_kernel_stats.csv' % sys.argv[1])[0]
rows = [r for r in csv.DictReader(open(f)) if any(k in r['Name'] for k in ('accumulate', 'finalize', 'level_finish', 'evidence'))]
print(sys.argv[1].ljust(8), '  '.join('%s x%s %.1f us' % (r['Name'].split('<')[-1][:24] if '<' in r['Name'] else r['Name'][7:22], r['Calls'], float(r['AverageNs']) / 1e3) for r in rows))
PY
  rm -rf gpurun_out/abo_$v
done
