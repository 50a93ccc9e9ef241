#!/bin/bash
# On the GPU box: the rocprofv3 counter passes behind profiles/rNN_pmc_counters.txt / rNN_pmc_traffic.json -- one run of the
# bench command per counter group (kernel trace only; --pmc never combined with other trace domains).
# usage: tools/collect_pmc.sh <tag> [bench arguments, default: configs[1]]
set -u
tag=${1:-run}; shift || true
root=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun exports GRAFT_REPO_ROOT)}
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
args=${*:---steps 5 --warmup 1 --ramp-ms 0 --no-other-configs}
: > "$out/counters.txt"
i=0
for group in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
             "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" \
             "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i + 1))
  bash "$root/tools/pmc_pass.sh" "$out/pass$i" "$group" -- python3 bench.py --no-cpu-baseline $args >> "$out/counters.txt" 2>> "$out/errors.txt"
  rm -rf "$out/pass$i"
done
wc -l "$out/counters.txt"
# configs[1] (the default arguments): the summary bench.py attaches to a run of the same (config, n, S)
if [ $# -eq 0 ]; then python3 "$root/tools/summarise_pmc.py" "$out/counters.txt" 3000 10000 > "$out/pmc_traffic.json"; fi
