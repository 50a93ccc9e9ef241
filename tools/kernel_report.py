#!/usr/bin/env python3
"""Registers / LDS / scratch / occupancy of every kernel of libdlagp_hip.so from hipcc's remarks (no GPU needed)."""
import os
import re
import subprocess
import sys

csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpy_dla_detection_amd", "csrc")
res = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                      "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", "dlagp_hip.hip"] + sys.argv[1:],
                     cwd=csrc, capture_output=True, text=True)
name, rows = None, {}
for line in res.stderr.splitlines():
    m = re.search(r"remark: +([^:]+): +(\S+)", line)
    if not m:
        continue
    key, val = m.group(1).strip(), m.group(2)
    if key == "Function Name":
        name = subprocess.run(["c++filt", val], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void dlagp::", "").replace("dlagp::", "")
        rows[name] = {}
    elif name:
        rows[name][key] = val
for name, r in rows.items():
    print("%-62s VGPR %-4s AGPR %-3s scratch %-5s occ %-2s LDS %-6s SGPR %s" % (
        name[:62], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"),
        r.get("LDS Size [bytes/block]"), r.get("SGPRs")))
