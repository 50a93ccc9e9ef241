#!/bin/bash
# On the GPU box: interleaved A/B of builds of libdlagp_hip.so over several bench configurations.
# usage: tools/ab_configs.sh <rounds> "<config> ..." <libA.so> <libB.so> ...
R=$1; CONFIGS=$2; shift 2
for i in $(seq $R); do
  for c in $CONFIGS; do
    for L in "$@"; do
      DLAGP_LIB=$PWD/$L python3 bench.py --config $c --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('config %s %-14s kernel_ms %.4f  ms/step %.4f  frac %.4f' % ('$c', '$L'.split('/')[-1], d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac']))"
    done
  done
done
