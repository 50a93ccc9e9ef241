#!/usr/bin/env python3
"""
The drop-in route's device efficiency from a rocprofv3 kernel-stats CSV of `tools/measure_route_a.py N` and that run's own
JSON line: kernel time per spectrum, its split by kernel family, and the algorithmic FLOP of the spectra over
the kernels' time and over the accumulate kernels' time, against the dense FP64 peak.
The null model's two kernels run on a second stream BESIDE the sweep's (dlagp_hip.hip, launch_null_side): the sum of the kernels'
durations then counts some instants twice; with the timeline of tools/route_a_timeline.py as third argument the device-busy
time (the union of the kernels' intervals) is reported as well, and the fraction of peak over it.
usage: tools/route_a_frac.py <kernel_stats.csv> <route_a.json> [route_a_timeline.json] > profiles/r06_route_a.json
"""
import csv
import json
import sys

PEAK = 78.6e12
rows = list(csv.DictReader(open(sys.argv[1])))
line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
# the tool runs the loop once over two spectra (warm-up) and once over all of them: every spectrum costs the same kernels
spectra = line["spectra"] + 2
# (finalize_rank20<.., false>: partial sums in the generic kernel's layout -- on this route only the null model's)
families = {"accumulate_mfma": "accumulate_mfma", "accumulate_generic": "null model (generic kernel + its finalisation, side stream)",
            "true, false>": "null model (generic kernel + its finalisation, side stream)",
            "false, false>": "null model (generic kernel + its finalisation, side stream)", "finalize": "finalisation",
            "level_": "level finish", "copy": "transfers (copy_segments + blit copies)", "sweep_init": "sweep_init",
            "interp_model": "set-up", "build_": "set-up", "fill_records": "set-up"}
split, total = {}, 0.0
for r in rows:
    ns = float(r["TotalDurationNs"])
    total += ns
    name = r["Name"]
    family = next((label for key, label in families.items() if key in name), "other")
    split[family] = split.get(family, 0.0) + ns
gflop = line["algorithmic_gflop_per_spectrum"]
out = dict(line)
out.update({
    "kernel_ms_per_spectrum": total / 1e6 / spectra,
    "kernel_us_per_spectrum_by_family": {k: round(v / 1e3 / spectra, 1) for k, v in sorted(split.items(), key=lambda kv: -kv[1])},
    "kernel_frac_of_fp64_peak": gflop * 1e9 / (total / 1e9 / spectra) / PEAK,
    "accumulate_frac_of_fp64_peak": gflop * 1e9 / (split.get("accumulate_mfma", 0.0) / 1e9 / spectra) / PEAK,
    "launches_per_spectrum": round(sum(int(r["Calls"]) for r in rows) / spectra, 1),
    "source": "rocprofv3 --kernel-trace --stats -- python3 tools/measure_route_a.py %d (wall figures: the same tool without the profiler)" % line["spectra"],
})
if len(sys.argv) > 3:
    timeline = json.load(open(sys.argv[3]))["median_us"]
    busy = timeline["device_busy_in_chain"]
    out.update({
        "device_busy_ms_per_spectrum": busy / 1e3,
        "busy_frac_of_fp64_peak": gflop * 1e9 / (busy / 1e6) / PEAK,
        "timeline_us_per_spectrum": {"host_between_spectra": timeline["host_gap_between_chains"],
                                     "first_kernel_to_first_accumulate": timeline["first_kernel_to_first_accumulate"],
                                     "chain_span": timeline["chain_span"], "device_idle_in_chain": timeline["device_idle_in_chain"]},
    })
print(json.dumps(out, indent=1))
