#!/usr/bin/env python3
"""
Where a one-spectrum call's wall time goes, from a rocprofv3 kernel trace of tools/measure_route_a.py: the kernels of one
spectrum form a chain (set-up, five sweeps' launches, downloads); between two spectra the device waits for the host.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/measure_route_a.py 256
    tools/route_a_timeline.py DIR/*/*_kernel_trace.csv

One JSON document: per spectrum (median over the chains) the chain's span, the device-busy time inside it (the union of the
kernels' intervals: kernels of two streams that overlap count once), the stretch from the chain's first kernel to its first
accumulate_mfma launch (set-up kernels and the host's time to issue the sweep), the idle time inside the chain by (kernel
before -> kernel after) pair, and the host's gap between two chains.
"""
import csv
import json
import re
import statistics
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"<.*", "", name)
    return name.replace("void ", "").strip()


def main():
    path = sys.argv[1]
    threshold = 20e3  # ns between an upload's copy_segments and the set-up kernel it feeds
    rows = []
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    # one chain per spectrum: it starts with the context's set-up, whose second kernel is interp_model (the upload's
    # copy_segments in front of it belongs to it when it is there)
    starts = []
    for i, (start, end, name) in enumerate(rows):
        if name.endswith("interp_model"):
            first = i - 1 if i and rows[i - 1][2].endswith("copy_segments") and start - rows[i - 1][1] < threshold else i
            starts.append(first)
    chains = [rows[a:b] for a, b in zip(starts, starts[1:] + [len(rows)])]
    # the warm-up spectra and the process start belong to no steady state: keep chains of the usual launch count
    usual = statistics.median(len(c) for c in chains)
    steady = [c for c in chains if abs(len(c) - usual) <= 4]
    spans, busies, idles, host_gaps, prefixes = [], [], [], [], []
    pair_idle = defaultdict(float)
    for k, c in enumerate(steady):
        first, last = c[0][0], max(e for _, e, _ in c)
        spans.append((last - first) / 1e3)
        sweeps = [start for start, _, name in c if "accumulate_mfma" in name]
        if sweeps:
            prefixes.append((sweeps[0] - first) / 1e3)
        covered, edge = 0.0, first
        previous = None
        for start, end, name in c:
            if start > edge:
                if previous is not None:
                    pair_idle[(previous, name)] += (start - edge) / 1e3
                edge = start
            if end > edge:
                covered += end - edge
                edge = end
            previous = name
        busies.append(covered / 1e3)
        idles.append(spans[-1] - busies[-1])
    ends = [(c[0][0], max(e for _, e, _ in c)) for c in chains]
    for (s0, e0), (s1, _) in zip(ends, ends[1:]):
        host_gaps.append((s1 - e0) / 1e3)
    n = len(steady)
    top = sorted(pair_idle.items(), key=lambda kv: -kv[1])[:14]
    out = {
        "chains": len(chains), "steady_chains": n, "launches_per_chain": usual,
        "median_us": {"chain_span": statistics.median(spans), "device_busy_in_chain": statistics.median(busies),
                      "device_idle_in_chain": statistics.median(idles), "host_gap_between_chains": statistics.median(host_gaps),
                      "first_kernel_to_first_accumulate": statistics.median(prefixes) if prefixes else None},
        "mean_us": {"chain_span": sum(spans) / n, "device_busy_in_chain": sum(busies) / n,
                    "device_idle_in_chain": sum(idles) / n, "host_gap_between_chains": sum(host_gaps) / max(len(host_gaps), 1)},
        "idle_in_chain_by_neighbours_us_per_spectrum": {"%s -> %s" % k: round(v / n, 2) for k, v in top},
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
