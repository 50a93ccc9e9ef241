#!/usr/bin/env python3
"""
profiles/r03_pmc_traffic.json from the per-kernel counter means tools/pmc_pass.sh printed (one rocprofv3 --pmc pass per
counter group of `bench.py --steps 5 --warmup 1 --ramp-ms 0 --no-cpu-baseline`), gfx950 read-side correction applied
(MI355X_MICROARCH.md, HBM: FETCH_SIZE counts 128-byte requests at 64 bytes for 16-byte-per-lane streaming reads).
usage: tools/summarise_pmc.py gpurun_out/pmc_r02_counters.txt [n = 3000] [S = 10000] > profiles/r03_pmc_traffic.json
"""
import json
import re
import sys

kernel, acc = None, {}
for line in open(sys.argv[1]):
    if not line.startswith(" "):
        kernel = line.strip()
        continue
    m = re.match(r"\s+(\S+)\s+mean\s+([0-9.]+)", line)
    if m and kernel:
        acc.setdefault(kernel, {})[m.group(1)] = float(m.group(2))
name = [k for k in acc if "accumulate_mfma" in k][0]
fin = [k for k in acc if "finalize_rank20" in k][0]
c, f = acc[name], acc[fin]
cycles = c["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
out = {
    # the workload the counters were collected on: bench.py attaches the traffic to a run only when all three match
    "config": 2, "n": int(sys.argv[2]) if len(sys.argv) > 2 else 3000, "S": int(sys.argv[3]) if len(sys.argv) > 3 else 10000,
    "kernel": "accumulate_mfma<3, true, false, 0, true, false> (3 lines, broadening, core patch; 14 full tiles + 2 small MFMAs per step)",
    "command": "rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 bench.py --steps 5 --warmup 1 --ramp-ms 0 --no-cpu-baseline  (one pass per counter group; tools/pmc_pass.sh)",
    "workload": "configs[1]: 1 sightline x 10000 samples, n=3000, k=20; 2500 work items (3 full pieces + remainder per block of 16 samples)",
    "FETCH_SIZE_KB_per_launch": c["FETCH_SIZE"],
    "WRITE_SIZE_KB_per_launch": c["WRITE_SIZE"],
    "gfx950_read_correction": 2.0,
    "hbm_bytes_per_launch": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0,
    "finalize_hbm_bytes_per_launch": (2.0 * f["FETCH_SIZE"] + f["WRITE_SIZE"]) * 1024.0,
    "TCC_HIT_sum": c["TCC_HIT_sum"], "TCC_MISS_sum": c["TCC_MISS_sum"],
    "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
    "SQ_INSTS_MFMA": c["SQ_INSTS_MFMA"], "SQ_INSTS_VALU": c["SQ_INSTS_VALU"],
    "non_mfma_valu_instructions": c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"],
    "SQ_INSTS_LDS": c["SQ_INSTS_LDS"], "SQ_INSTS_SALU": c["SQ_INSTS_SALU"],
    "SQ_VALU_MFMA_BUSY_CYCLES": c["SQ_VALU_MFMA_BUSY_CYCLES"],
    "SQ_VALU_MFMA_COEXEC_CYCLES": c.get("SQ_VALU_MFMA_COEXEC_CYCLES"),
    "SQ_WAVES": c["SQ_WAVES"], "SQ_WAVE_CYCLES_quad": c["SQ_WAVE_CYCLES"],
    "SQ_WAIT_ANY_quad": c.get("SQ_WAIT_ANY"), "SQ_WAIT_INST_ANY_quad": c.get("SQ_WAIT_INST_ANY"),
    "SQ_ACTIVE_INST_VALU_quad": c.get("SQ_ACTIVE_INST_VALU"),
    "GRBM_GUI_ACTIVE_per_xcd": cycles,
    "mfma_busy_fraction_of_kernel_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cycles,
    "wave_slot_occupancy": c["SQ_WAVE_CYCLES"] * 4.0 / 2048.0 / cycles,
    "note": "MFMA busy = 6.5625 M x 64 + 0.9375 M x 16 cycles over 1024 SIMDs (13.92 GFLOP executed for 13.8 algorithmic); kernel cycles from GRBM_GUI_ACTIVE / 8; the counter "
            "passes run cold (no clock ramp), so cycles, not microseconds, are the comparable quantity",
}
print(json.dumps(out, indent=1))
