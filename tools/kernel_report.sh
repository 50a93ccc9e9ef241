#!/bin/bash
# Per-kernel register / LDS / scratch / occupancy table from hipcc's resource-usage remarks.
cd "$(dirname "$0")/../gpy_dla_detection_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage \
    -c -o /dev/null dlagp_hip.hip 2>&1 | python3 -c '
import re, sys
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark: .*?(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    key, val = m.group(1), m.group(2)
    if key == "Function Name":
        cur = {"name": val}; rows.append(cur)
    elif cur is not None:
        cur[key] = val
print("%-70s %5s %5s %5s %8s %5s %7s %7s" % ("kernel", "SGPR", "VGPR", "AGPR", "scratch", "occ", "vspill", "LDS"))
for r in rows:
    print("%-70s %5s %5s %5s %8s %5s %7s %7s" % (r["name"][:70], r.get("TotalSGPRs"), r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("VGPRs Spill"), r.get("LDS Size [bytes/block]")))
'
