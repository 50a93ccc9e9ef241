#!/usr/bin/env python3
"""
How busy the device was: reads a rocprofv3 `*_kernel_trace.csv`, takes the last `--window` seconds of it (the timed
run of tools/measure_catalogue.py comes last) and prints the union of the kernel intervals, the sum of the kernel
durations (more than the union when kernels of two streams overlap), the time per kernel name and the largest gaps.
usage: tools/trace_busy.py <kernel_trace.csv> [--window SECONDS]
"""
import csv
import json
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    window = float(sys.argv[sys.argv.index("--window") + 1]) if "--window" in sys.argv else None
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    end = max(r[1] for r in rows)
    begin = rows[0][0] if window is None else end - int(window * 1e9)
    rows = [r for r in rows if r[0] >= begin]
    begin = rows[0][0]
    union, cursor, gaps = 0, begin, []
    overlapped = 0  # time with at least two kernels resident
    events = []
    around = defaultdict(lambda: [0, 0.0])  # (kernel before the gap, kernel after it) -> count, total
    last_name = "-"
    for s, e, name in rows:
        events.append((s, 1))
        events.append((e, -1))
        if s > cursor:
            gaps.append((s - cursor, cursor - begin))
            key = (last_name.split("(")[0][-34:], name.split("(")[0][-34:])
            around[key][0] += 1
            around[key][1] += (s - cursor) / 1e6
            cursor = s
        if e >= cursor:
            last_name = name
        if e > cursor:
            union += e - cursor
            cursor = e
    events.sort()
    depth, last = 0, begin
    for t, d in events:
        if depth >= 2:
            overlapped += t - last
        depth += d
        last = t
    by_name = defaultdict(lambda: [0, 0])
    for s, e, name in rows:
        short = name.split("(")[0][:60]
        by_name[short][0] += e - s
        by_name[short][1] += 1
    span = end - begin
    out = {
        "span_ms": span / 1e6, "busy_ms": union / 1e6, "busy_frac": union / span,
        "two_or_more_kernels_ms": overlapped / 1e6,
        "sum_of_durations_ms": sum(e - s for s, e, _ in rows) / 1e6,
        "largest_gaps_ms_at_ms": [(round(g / 1e6, 3), round(at / 1e6, 1)) for g, at in sorted(gaps, reverse=True)[:8]],
        "gaps_over_50us": sum(1 for g, _ in gaps if g > 50e3), "gap_total_ms": sum(g for g, _ in gaps) / 1e6,
        "gap_ms_by_size": {label: round(sum(g for g, _ in gaps if lo <= g < hi) / 1e6, 2) for label, lo, hi in
                           (("<20us", 0, 20e3), ("20-100us", 20e3, 100e3), ("0.1-0.5ms", 100e3, 500e3),
                            ("0.5-2ms", 500e3, 2e6), (">2ms", 2e6, 1e18))},
        "gap_ms_between": {"%s -> %s" % k: [v[0], round(v[1], 2)] for k, v in
                           sorted(around.items(), key=lambda kv: -kv[1][1])[:12]},
        "kernels": {k: {"ms": round(v[0] / 1e6, 2), "calls": v[1]} for k, v in sorted(by_name.items(), key=lambda kv: -kv[1][0])[:12]},
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
