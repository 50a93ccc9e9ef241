#!/bin/bash
# On the GPU box: interleaved A/B of builds of libdlagp_hip.so on the same device, sustained clocks.
# usage: tools/ab_bench.sh <rounds> <libA.so> <libB.so> ...
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    DLAGP_LIB=$PWD/$L python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-14s kernel_ms %.4f  ms/step %.4f  frac %.4f' % ('$L'.split('/')[-1], d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac']))"
  done
done
