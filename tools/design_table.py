#!/usr/bin/env python3
"""Markdown tables of DESIGN.md section 7 from a collection directory (tools/collect_profiles.sh <tag>): usage design_table.py profiles r05"""
import csv
import json
import os
import sys

root, tag = sys.argv[1], sys.argv[2]


def load(name):
    path = os.path.join(root, "%s_%s" % (tag, name)) if root.endswith("profiles") else os.path.join(root, name)
    text = open(path).read().strip()
    try:
        return json.loads(text)  # one JSON document, possibly indented
    except ValueError:
        return json.loads(text.splitlines()[-1])  # a tool's output with its one JSON line last


def kernel_avg(csv_name, needle):
    path = os.path.join(root, "%s_%s" % (tag, csv_name)) if root.endswith("profiles") else os.path.join(root, csv_name)
    for r in csv.DictReader(open(path)):
        if needle in r["Name"]:
            return float(r["AverageNs"]) / 1e3, int(r["Calls"])
    return None, 0


b = load("bench.json")
r = b["roofline"]
print("| quantity | round %s | file |" % tag.lstrip("r0"))
print("|---|---|---|")
print("| step, configs[1] | %.4f ms → **%.1f M evals/s**%s | `%s_bench.json` |"
      % (b["ms_per_step"], b["value"] / 1e6, (" (`step_frac` %.1f %% of FP64 peak)" % (100 * r["step_frac"])) if "step_frac" in r else "", tag))
avg, calls = kernel_avg("kernel_stats.csv", "accumulate_mfma<3, true, false, 0, 1, false, false")
print("| accumulate_mfma, configs[1] | %.4f ms by HIP events → %.1f TFLOP/s = **%.1f %%** of 78.6 (`frac_cold` %.1f %%); %.4f ms avg over %d launches under rocprofv3 | `%s_bench.json`, `%s_kernel_stats.csv` |"
      % (r["kernel_ms"], r["achieved"], 100 * r["frac"], 100 * r["frac_cold"], (avg or 0) / 1e3, calls, tag, tag))
fin, _ = kernel_avg("kernel_stats.csv", "finalize_rank20")
ev, _ = kernel_avg("kernel_stats.csv", "level_evidence")
if ev:
    print("| finalize_rank20 / level_evidence | %.1f / %.1f µs | `%s_kernel_stats.csv` |" % (fin or 0, ev, tag))
else:  # round 6 on: the step's evidence comes out of the finalisation kernel's tail (dlagp_eval_evidence_device)
    print("| finalize_rank20 with the level evidence in its tail (no level_evidence launch) | %.1f µs | `%s_kernel_stats.csv` |" % (fin or 0, tag))
for c, what in ((3, "64 sightlines × {sub-DLA, DLA(1)}: one pair launch"), (4, "one rank's 128 of 1024: null + sub-DLA + DLA(1)"),
                (5, "one rank's 32 of 256: DLA(1..4), cached levels")):
    x = load("bench_config%d.json" % c)
    o = b["other_configs"][str(c)]
    print("| config %d (%s) | %.1f ms, **%.1f M evals/s**%s, accumulate %.1f %% — inside the default run: %.1f ms | `%s_bench_config%d.json` |"
          % (c, what, x["ms_per_step"], x["value"] / 1e6, (", **%.0f spectra/s**" % x["spectra_per_s"]) if x.get("spectra_per_s") else "",
             100 * x["roofline"]["frac"], o["ms_per_step"], tag, c))
s = b["strong_n1"]
print("| by-sample split at N = 1 | %.4f ms, %.1f M evals/s | `%s_bench.json` |" % (s["ms_per_step"], s["evals_per_s"] / 1e6, tag))
for name, label in (("catalogue_pipeline.json", "z_qso uniform in [2.2, 3.4]"), ("catalogue_pipeline_dr12q.json", "DR12Q-like z_qso")):
    x = load(name)
    idle = x.get("kernel_stream_idle")
    runs = x.get("spectra_per_s_runs")
    rate = x.get("spectra_per_s_median", x["spectra_per_s"])
    print("| catalogue driver, 2048 spectra, %s (n median %.0f, mean %.0f) | **%.0f spectra/s**%s, %.1f M evals/s, whole run %.1f %% of FP64 peak%s%s | `%s_%s` |"
          % (label, x["kept_pixels"]["median"], x["kept_pixels"]["mean"], rate,
             " (median of %s)" % " / ".join("%.0f" % r for r in runs) if runs else "",
             rate * (1 + 5 * x["S"]) / 1e6, 100 * x["whole_run_frac_of_fp64_peak"] * rate / x["spectra_per_s"],
             "; last run: accumulate kernels %.1f %% of the wall time at %.1f %% of peak" % (100 * x["accumulate_share_of_run"], 100 * x["accumulate_frac_of_fp64_peak"]) if "accumulate_share_of_run" in x else "",
             ", kernel stream found empty at %.1f %% of %d polls" % (100 * idle["share"], idle["polls"]) if idle else "", tag, name))
try:
    x = load("catalogue_pipeline_dr12q_8192.json")
    print("| catalogue driver, 8192 spectra, DR12Q-like z_qso | **%.0f spectra/s**, whole run %.1f %% of FP64 peak | `%s_catalogue_pipeline_dr12q_8192.json` |"
          % (x["spectra_per_s"], 100 * x["whole_run_frac_of_fp64_peak"], tag))
    x = load("bench_pixels_685.json")
    print("| one sightline of DR12Q's median length (`bench.py --pixels 685`: n = %d) | step %.4f ms → **%.1f M evals/s**; accumulate %.4f ms = %.1f %% of peak | `%s_bench_pixels_685.json` |"
          % (round(x["roofline"]["flop_per_eval"] / 460), x["ms_per_step"], x["value"] / 1e6, x["roofline"]["kernel_ms"], 100 * x["roofline"]["frac"], tag))
except (OSError, KeyError):
    pass
x = load("route_a.json")
if "kernel_ms_per_spectrum" in x:  # round 6 on: with the kernels' own time from the rocprofv3 run of the same command
    fam = x["kernel_us_per_spectrum_by_family"]
    null = next((v for k, v in fam.items() if k.startswith("null model")), 0)
    if "device_busy_ms_per_spectrum" in x:  # ... and the kernels' timeline (two streams: summed durations count overlaps twice)
        t = x["timeline_us_per_spectrum"]
        busy = ("device busy %.3f ms per spectrum = **%.1f %%** of FP64 peak over that time (kernels summed %.3f ms: the null model's two run beside the sweep's); "
                % (x["device_busy_ms_per_spectrum"], 100 * x["busy_frac_of_fp64_peak"], x["kernel_ms_per_spectrum"]))
        where = ("; of the wall time %.0f µs host between two spectra, %.0f µs from a spectrum's first kernel to its first accumulate launch"
                 % (t["host_between_spectra"], t["first_kernel_to_first_accumulate"]))
    else:
        busy = "kernels %.3f ms per spectrum = **%.1f %%** of FP64 peak over their own time " % (x["kernel_ms_per_spectrum"], 100 * x["kernel_frac_of_fp64_peak"])
        where = ""
    print("| route A (reference driver loop, imports swapped), 256 spectra, n median %.0f | **%.2f ms per spectrum** (%.0f spectra/s; three runs %s); %s(accumulate %.0f µs at %.1f %%, level finish %.0f, finalisation %.0f, transfers %.0f, set-up %.0f, null %.0f); %.0f launches per spectrum%s | `%s_route_a.json`, `%s_kernel_stats_route_a.csv`, `%s_route_a_timeline.json` |"
          % (x["kept_pixels"]["median"], x["ms_per_spectrum"], x["spectra_per_s"], " / ".join("%.3f" % v for v in x.get("ms_per_spectrum_runs", [])),
             busy, fam.get("accumulate_mfma", 0), 100 * x["accumulate_frac_of_fp64_peak"], fam.get("level finish", 0), fam.get("finalisation", 0),
             fam.get("transfers (copy_segments + blit copies)", 0), fam.get("set-up", 0), null,
             x["launches_per_spectrum"], where, tag, tag, tag))
else:
    print("| route A (reference driver loop, imports swapped), 256 spectra | **%.2f ms per spectrum** (%.0f spectra/s) | `%s_route_a.json` |"
          % (x["ms_per_spectrum"], x["spectra_per_s"], tag))
cb = b["cpu_baseline"]
print("| CPU baseline (oracle, %d pinned processes × 1 BLAS thread) | **%.1f k evals/s** (chunks %s; ramp %s); 1 process %.1f k; default threads %.1f k on %s threads | `%s_bench.json` |"
      % (cb["cores"], cb["value"] / 1e3, " / ".join("%.1f" % (v / 1e3) for v in cb["chunk_rates"]),
         " / ".join("%.1f" % (v / 1e3) for v in cb.get("untimed_ramp_chunk_rates", [cb.get("untimed_first_chunk_rate", 0)])),
         cb["variants"]["i_one_process_one_blas_thread"]["value"] / 1e3,
         cb["variants"]["ii_one_process_default_blas_threads"].get("value", 0) / 1e3,
         cb["variants"]["ii_one_process_default_blas_threads"].get("blas_threads_used"), tag))
print()
print("**Efficiency against the sightline length** (batch of 32 × 10 000 samples, one stream; `frac` = accumulate kernels, `step` = wall "
      "time incl. finalisation and level_finish; `%s_frac_vs_n.json`, `%s_kernel_stats_n*.csv`):\n" % (tag, tag))
print("| n_u (n kept) | sub-DLA sweep | sub-DLA + DLA(1), one pair launch | level 1 + store | cached level | model selection, 5 levels: step | prologue / epilogue cycles per wave |")
print("|---|---|---|---|---|---|---|")
for d in load("frac_vs_n.json"):
    st = d.get("stamps_cycles_per_wave", {})
    print("| %d (%.0f) | %.3f ms, frac %.3f, step %.3f | %.3f ms, frac %.3f, step %.3f | %.3f ms, %.3f | %.3f ms, %.3f | %.2f ms, **%.3f** | %.0f k / %.0f k of %.0f k |"
          % (d["pixels_unmasked"], d["n_kept_mean"], d["sub_dla"]["ms_per_launch"], d["sub_dla"]["frac"], d["sub_dla"]["step_frac"],
             d["sub_dla_plus_dla1"]["ms_per_launch"], d["sub_dla_plus_dla1"]["frac"], d["sub_dla_plus_dla1"]["step_frac"],
             d["level1_with_store"]["ms_per_launch"], d["level1_with_store"]["frac"], d["cached_level"]["ms_per_launch"],
             d["cached_level"]["frac"], d["model_selection_5_levels"]["step_wall_ms"], d["model_selection_5_levels"]["step_frac"],
             st.get("prologue", 0) / 1e3, st.get("epilogue", 0) / 1e3, (st.get("passes_and_epilogue", 0) + st.get("prologue", 0)) / 1e3))
try:
    d = load("frac_ragged_mix.json")
    print("| ragged: n_u uniform in %d…%d (mean n %.0f) | %.3f ms, frac %.3f, step %.3f | %.3f ms, frac %.3f, step %.3f | %.3f ms, %.3f | %.3f ms, %.3f | %.2f ms, **%.3f** | |"
          % (round(d["pixels_unmasked"] * (1 - d["spread"])), round(d["pixels_unmasked"] * (1 + d["spread"])), d["n_kept_mean"],
             d["sub_dla"]["ms_per_launch"], d["sub_dla"]["frac"], d["sub_dla"]["step_frac"], d["sub_dla_plus_dla1"]["ms_per_launch"],
             d["sub_dla_plus_dla1"]["frac"], d["sub_dla_plus_dla1"]["step_frac"], d["level1_with_store"]["ms_per_launch"],
             d["level1_with_store"]["frac"], d["cached_level"]["ms_per_launch"], d["cached_level"]["frac"],
             d["model_selection_5_levels"]["step_wall_ms"], d["model_selection_5_levels"]["step_frac"]))
except (OSError, KeyError):
    pass
