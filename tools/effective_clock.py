#!/usr/bin/env python3
"""Effective shader clock of the accumulate kernel: GRBM_GUI_ACTIVE cycles / kernel duration, per launch,
from one rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace run (tools/pmc_pass.sh output directory)."""
import csv, glob, sys
d = sys.argv[1]
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "accumulate" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
rows = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "accumulate" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), dur[r["Dispatch_Id"]]))
rows.sort()
tail = rows[len(rows) // 2:]
for name, part in (("first 5", rows[:5]), ("second half", tail)):
    if part:
        cyc = sum(p[1] for p in part) / len(part)
        ns = sum(p[2] for p in part) / len(part)
        print("%-12s launches %4d: GRBM_GUI_ACTIVE %.0f cycles, duration %.1f us -> %.3f GHz" % (name, len(part), cyc, ns / 1e3, cyc / ns))
