#!/bin/bash
# On the GPU box: the frac-vs-n curve (tools/measure_frac_vs_n.py) for n_u in {300, 685, 1250, 3061}, once plainly (with the
# per-phase stamps) and once under rocprofv3 --kernel-trace --stats.   usage: tools/collect_frac_vs_n.sh <tag> [extra args]
set -u
tag=${1:-r5}; shift
root=${GRAFT_REPO_ROOT:?run this on the GPU box}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root"
for n in 300 685 1250 3061; do
  python3 tools/measure_frac_vs_n.py --pixels $n --stamps "$@" > $out/frac_n$n.json 2> $out/frac_n$n.err || exit 1
  echo "n=$n done"
done
for n in 300 685 1250 3061; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_n$n -- python3 tools/measure_frac_vs_n.py --pixels $n "$@" > $out/frac_n${n}_under_rocprof.json 2> $out/prof_n$n.err || exit 1
  cp $out/prof_n$n/*/*_kernel_stats.csv $out/kernel_stats_n$n.csv
  rm -rf $out/prof_n$n
  echo "rocprof n=$n done"
done
