#!/bin/bash
# On the GPU box: per-kernel averages (rocprofv3 kernel stats) of a catalogue run for several builds of the library.
# usage: tools/ab_catalogue_kernels.sh "<measure_catalogue args>" <libA.so> <libB.so> ...
ARGS=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for L in "$@"; do
  tag=$(basename $L .so)
  DLAGP_LIB=$PWD/$L rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abc_$tag -- python3 tools/measure_catalogue.py $ARGS > gpurun_out/abc_$tag.log 2>&1
  python3 - $tag <<'PY'
import csv, glob, sys
f = glob.glob('gpurun_out/abc_%s/*/*_kernel_stats.csv' % sys.argv[1])[0]
rows = [r for r in csv.DictReader(open(f)) if any(k in r['Name'] for k in ('accumulate', 'finalize', 'level_finish'))]
print(sys.argv[1].ljust(8), '  '.join('%s x%s %.1f us' % (r['Name'].split('<')[-1][:24] if '<' in r['Name'] else r['Name'][7:22], r['Calls'], float(r['AverageNs']) / 1e3) for r in rows))
PY
  grep spectra_per_s gpurun_out/abc_$tag.log | cut -c1-130
  rm -rf gpurun_out/abc_$tag
done
