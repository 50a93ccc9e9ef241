#!/usr/bin/env python3
"""
From a rocprofv3 `*_kernel_trace.csv` of `python3 bench.py`: the average duration of the accumulate kernel over the LAST
`steps` launches (bench.py's timed region comes last; the 25 cold steps, the clock ramp and the warm-up precede it), next
to the average over all launches that `--stats` reports.  usage: tools/timed_region_average.py <kernel_trace.csv> [steps]
"""
import csv
import json
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if "accumulate_mfma" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dur = [e - s for s, e in rows]
timed = dur[-steps:]
print(json.dumps({
    "kernel": "accumulate_mfma", "launches": len(dur), "average_all_launches_us": sum(dur) / len(dur) / 1e3,
    "timed_region_launches": len(timed), "average_timed_region_us": sum(timed) / len(timed) / 1e3,
    "min_timed_region_us": min(timed) / 1e3, "max_timed_region_us": max(timed) / 1e3,
}))
