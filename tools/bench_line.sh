#!/bin/bash
# one summary line of a bench.py run: tools/bench_line.sh <label> [bench.py arguments]
label=$1; shift
python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('%-28s ms/step %.4f  kernel_ms %.4f  frac %.3f  evals/s %.3e' % ('$label', d['ms_per_step'], r['kernel_ms'], r['frac'], d['value']))"
