#!/bin/bash
# On the GPU box: A/B of one library option over bench configurations.  usage: tools/ab_option.sh <rounds> "<configs>" <name> <v0> <v1> ...
R=$1; CONFIGS=$2; NAME=$3; shift 3
for i in $(seq $R); do for c in $CONFIGS; do for v in "$@"; do
  python3 bench.py --config $c --no-cpu-baseline --option $NAME=$v 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $c  $NAME=$v  kernel_ms %.4f  ms/step %.4f  frac %.4f' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac']))"
done; done; done
