#!/bin/bash
# usage (on the GPU box): tools/pmc_pass.sh <outdir> "<COUNTER ...>" -- <program> [args]
# One rocprofv3 counter pass (own run, kernel trace only), then a per-kernel mean of every counter.
set -e
out=$1; counters=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run this on the GPU box}"
mkdir -p "$out"
rocprofv3 --pmc $counters --kernel-trace --output-format csv -d "$out" -- "$@" > "$out/run.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "dlagp" not in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        print("    %-32s mean %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
