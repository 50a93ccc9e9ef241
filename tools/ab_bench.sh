#!/bin/bash
# On the GPU box: interleaved A/B of two builds of libdlagp_hip.so on the same device.
# usage: tools/ab_bench.sh <libA.so> <libB.so> [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in "$A" "$B"; do
    DLAGP_LIB=$L python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1], 'kernel_ms %.4f  ms/step %.4f' % (d['roofline']['kernel_ms'], d['ms_per_step']))"
  done
done
