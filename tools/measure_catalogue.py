#!/usr/bin/env python3
"""
End-to-end catalogue throughput on one GPU: raw spectra -> run_bayes_select.process_qso ->
the reference's output arrays.  Synthetic SDSS-like spectra (4600 pixels at 1e-4 dex from 3631 A),
S = 10 000 QMC samples, max_dlas = 4.  Prints one JSON object with spectra/s and a phase split.

    python3 tools/measure_catalogue.py [spectra] [S] [batch size] [mix] [sweeps-only]

With a fifth argument `sweeps-only` the batches (at most 12) are prepared first and only their model selections are timed,
one after the other on one thread: what the device does with the pipeline's work when nothing else shares it -- the
bound the pipeline's spectra/s is to be read against.

The kept pixels n of a spectrum follow z_qso through the modelled window 911.75-1215.75 A (reference
set_parameters.py:33-34, null_gp.py:139-152): the window is entered at z_qso ~ 2.0 and fully covered, n = 1250, from
z_qso ~ 2.98.  Two redshift mixes:
  uniform  z_qso uniform in [2.2, 3.4]: n = 300 ... 1250, mean ~ 920 (rounds 1-4; their docstring said "n ~ 1000-1250",
           which only holds above z_qso ~ 2.8)
  dr12q    z_qso = 2.15 + Gamma(2, 0.22), the DR12Q-like spread of tests/test_host_logic.py
           (test_cost_balanced_partition_of_a_redshift_ordered_catalogue): n = 250 ... 1250, median ~ 685
The output lists the n statistics of the spectra it ran on.
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch

torch.cuda.init()
from gpy_dla_detection_amd.run_bayes_select import LearnedModel, process_qso  # noqa: E402
from gpy_dla_detection_amd.samples import DLASamples, SubDLASamples, halton  # noqa: E402
from gpy_dla_detection_amd.set_parameters import Parameters  # noqa: E402


class FixedPrior:
    def less_ind(self, z_qso):
        return 731.0, 9265.0


from gpy_dla_detection_amd.synthetic import make_catalogue as make_inputs  # noqa: E402,F401  (tools and tests import it from here)


def sweeps_only(spectra, z_qsos, params, dla, sub, learned, batch_size, S):
    from gpy_dla_detection_amd.null_gp import prepare_data
    from gpy_dla_detection_amd.run_bayes_select import DeviceEngine

    engine = DeviceEngine(learned, params)
    nhi_dla = np.asarray(dla.nhi_samples, dtype=np.float64)
    nhi_sub = np.asarray(sub.nhi_samples, dtype=np.float64)
    min_sep = params.kms_to_z(3000.0)
    prepared = []
    for start in range(0, min(len(spectra), 12 * batch_size), batch_size):
        sightlines, z_dla, z_lls = [], [], []
        for q in range(start, min(start + batch_size, len(spectra))):
            wavelengths, flux, noise_variance, pixel_mask = spectra[q]
            rest = params.emitted_wavelengths(wavelengths, z_qsos[q])
            st = prepare_data(params, rest, flux, noise_variance, pixel_mask, z_qsos[q], normalize=True)
            sightlines.append(engine.sightline(st, z_qsos[q], True))
            z_dla.append(dla.sample_z_dlas(st["this_wavelengths"], z_qsos[q]))
            z_lls.append(sub.sample_z_lls(st["this_wavelengths"], z_qsos[q]))
        prepared.append((sightlines, engine.batch(sightlines), np.stack(z_dla), np.stack(z_lls)))
    from gpy_dla_detection_amd.sightline import set_option, timing_read

    times = []
    for rep in range(3):  # the first round allocates the workspaces
        if rep == 2:
            set_option("timing", 1)  # event pairs around the accumulate launches of the last round
            timing_read(True)
        t0 = time.perf_counter()
        for sightlines, batch, z_dla, z_lls in prepared:
            engine.model_evidences(batch, z_lls, nhi_sub, z_dla, nhi_dla, 4, min_sep, 0)
        times.append(time.perf_counter() - t0)
    accumulate_ms, launches = timing_read(True)
    set_option("timing", 0)
    count = sum(len(p[0]) for p in prepared)
    for sightlines, batch, _, _ in prepared:
        engine.release(batch, sightlines)
    engine.close()
    return {"batches": len(prepared), "spectra": count, "seconds_per_round": times, "spectra_per_s": count / min(times[1:]),
            "accumulate_ms_last_round": accumulate_ms, "accumulate_launches_last_round": launches}


def main():
    num = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    batch_size = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    mix = sys.argv[4] if len(sys.argv) > 4 else "uniform"
    if mix not in ("uniform", "dr12q"):
        raise SystemExit("mix must be uniform or dr12q")
    params = Parameters(num_dla_samples=S)
    u = halton(S, 3)
    dla = DLASamples(params, None, halton(S, 2), 20.0 + 3.0 * u)
    sub = SubDLASamples(params, None, halton(S, 2), 19.5 + 0.5 * u, Z_dla=1.0, Z_lls=0.25)
    learned, spectra, z_qsos = make_inputs(num, mix=mix)
    import functools

    from gpy_dla_detection_amd import run_bayes_select as _rbs
    from gpy_dla_detection_amd.sightline import set_options_from_env

    options = set_options_from_env()
    native = os.environ.get("DLAGP_NATIVE_PREPARE", "1") != "0"  # A/B: the per-spectrum Python preparation
    global process_qso
    process_qso = functools.partial(_rbs.process_qso, native_prepare=native)
    from gpy_dla_detection_amd.run_bayes_select import kept_pixel_counts

    kept = kept_pixel_counts(params, spectra, z_qsos)
    if len(sys.argv) > 5 and sys.argv[5] == "sweeps-only":
        out = sweeps_only(spectra, z_qsos, params, dla, sub, learned, batch_size, S)
        used = kept[: out["spectra"]]
        out["accumulate_frac_of_fp64_peak"] = float(used.sum()) * (20 * 21 + 40) * 5 * S / (out["accumulate_ms_last_round"] * 1e-3) / 78.6e12
        out.update({"mix": mix, "S": S, "batch_size": batch_size, "kept_pixels_mean": float(used.mean()),
                    "frac_of_fp64_peak": float(used.sum()) * (20 * 21 + 40) * 5 * S / (out["spectra"] / out["spectra_per_s"]) / 78.6e12})
        print(json.dumps(out))
        return
    for item in os.environ.get("DLAGP_OPTIONS", "").split(","):  # e.g. DLAGP_OPTIONS=shared_work_stream=0
        if item:
            from gpy_dla_detection_amd.sightline import set_option

            set_option(item.split("=")[0], int(item.split("=")[1]))
    if os.environ.get("DLAGP_THREE_CALLS"):  # the driver before dlagp_batch_model_evidences: nulls + two sweeps
        from gpy_dla_detection_amd import run_bayes_select

        del run_bayes_select.DeviceEngine.model_evidences
    # warm-up through the pipeline itself: every workspace it keeps in flight gets its buffers (pinned staging, GBs of cache)
    warm = min(num, 6 * batch_size)
    process_qso(spectra[:warm], z_qsos[:warm], params, FixedPrior(), dla, sub, learned, batch_size=batch_size)
    from gpy_dla_detection_amd.sightline import get_option

    # DLAGP_RUNS=n: n - 1 plain timed runs in front of the (instrumented) last one; single runs scatter by +-4 %
    earlier_runs = []
    for _ in range(int(os.environ.get("DLAGP_RUNS", "1")) - 1):
        t0 = time.perf_counter()
        process_qso(spectra, z_qsos, params, FixedPrior(), dla, sub, learned, batch_size=batch_size)
        earlier_runs.append(num / (time.perf_counter() - t0))
    allocs_before = (get_option("big_allocs"), get_option("big_frees"))
    free_before = torch.cuda.mem_get_info()[0]
    polls = [0, 0]
    idle_at, calls = [], []
    if os.environ.get("DLAGP_POLL_IDLE"):  # a side thread asks every 0.5 ms whether the kernel stream has run dry
        import threading

        stop_polling = threading.Event()

        def poll():
            torch.cuda.set_device(0)
            while not stop_polling.is_set():
                polls[0] += 1
                if get_option("work_stream_idle"):
                    polls[1] += 1
                    idle_at.append(time.perf_counter())
                time.sleep(0.0005)

        if os.environ.get("DLAGP_POLL_IDLE") == "who":
            # ... and what every worker thread was inside at those instants (the engine's calls, timed)
            from gpy_dla_detection_amd import run_bayes_select as rbs

            def timed(name, fn):
                def wrapper(*args, **kwargs):
                    t0 = time.perf_counter()
                    try:
                        return fn(*args, **kwargs)
                    finally:
                        calls.append((name, threading.get_ident(), t0, time.perf_counter()))
                return wrapper

            for name in ("sightline", "batch", "model_evidences", "release"):
                setattr(rbs.DeviceEngine, name, timed(name, getattr(rbs.DeviceEngine, name)))
            rbs.prepare_data = timed("prepare_data", rbs.prepare_data)
            rbs.occupation_log_priors = timed("log_priors", rbs.occupation_log_priors)

        poller = threading.Thread(target=poll, daemon=True)
        poller.start()
    if os.environ.get("DLAGP_TIME_ACCUMULATE"):  # event pairs around every accumulate launch of the timed run
        from gpy_dla_detection_amd.sightline import set_option, timing_read

        set_option("timing", 1)
        timing_read(True)
    t0 = time.perf_counter()
    res = process_qso(spectra, z_qsos, params, FixedPrior(), dla, sub, learned, batch_size=batch_size)
    t = time.perf_counter() - t0
    accumulate = timing_read(True) if os.environ.get("DLAGP_TIME_ACCUMULATE") else None
    if polls[0]:
        stop_polling.set()
        poller.join()
    free_after = torch.cuda.mem_get_info()[0]
    out = {
        "spectra": num, "S": S, "max_dlas": 4, "batch_size": batch_size, "seconds": t, "mix": mix,
        "native_prepare": native, "options": options,
        "kept_pixels": {"min": int(kept.min()), "median": float(np.median(kept)), "mean": float(kept.mean()), "max": int(kept.max())},
        # algorithmic FLOP of the run (n k (k + 1) + 2 n k per eval, 5 S evals per spectrum) / wall time / peak
        "whole_run_frac_of_fp64_peak": float(kept.sum()) * (20 * 21 + 40) * 5 * S / t / 78.6e12,
        "spectra_per_s": num / t, "evals_per_s": num * (1 + 5 * S) / t,
        "argmax_counts": np.bincount(np.argmax(res["model_posteriors"], axis=1), minlength=6).tolist(),
        # device memory held by the library's pools after the warm-up, and what the timed run added to it
        "hbm_in_use_gb": round((torch.cuda.mem_get_info()[1] - free_after) / 2 ** 30, 2),
        "hbm_growth_during_run_gb": round((free_before - free_after) / 2 ** 30, 3),
        "workspace_allocs_frees_during_run": [get_option("big_allocs") - allocs_before[0], get_option("big_frees") - allocs_before[1]],
    }
    if earlier_runs:
        out["spectra_per_s_runs"] = earlier_runs + [num / t]
        out["spectra_per_s_median"] = float(np.median(out["spectra_per_s_runs"]))
    if accumulate:
        # the accumulate kernels' own time inside the pipeline: against `seconds` it says how much of the run they fill,
        # against the same sweeps run alone (sweeps-only) how much the work beside them slows them down
        out["accumulate_ms"], out["accumulate_launches"] = accumulate
        out["accumulate_frac_of_fp64_peak"] = float(kept.sum()) * (20 * 21 + 40) * 5 * S / (accumulate[0] * 1e-3) / 78.6e12
        out["accumulate_share_of_run"] = accumulate[0] * 1e-3 / t
    if polls[0]:
        out["kernel_stream_idle"] = {"polls": polls[0], "found_idle": polls[1], "share": polls[1] / polls[0]}
    if calls:
        from collections import Counter

        names = {"sightline": "p", "prepare_data": "p", "batch": "B", "model_evidences": "S", "release": "r", "log_priors": "f"}
        states = Counter()
        for t in idle_at:
            inside = sorted(names[n] for n, _, a, b in calls if a <= t <= b)
            states["".join(inside) or "-"] += 1
        # p: a context being made, B: a batch object being made, S: a thread inside the library's model selection call,
        # f / r: finishing; e.g. "SS" = two sweep threads inside the call, nobody preparing
        out["threads_inside_at_idle_instants"] = dict(states.most_common(12))
        sweeps = sorted((a, b) for n, _, a, b in calls if n == "model_evidences")
        out["model_evidences_ms"] = {"mean": 1e3 * float(np.mean([b - a for a, b in sweeps])), "calls": len(sweeps)}
    if os.environ.get("DLAGP_PROFILE_THREADS"):
        # where the HOST's time goes, worker threads included: one cProfile per thread on the thread's own CPU clock
        # (waiting for the interpreter lock or inside a blocking call does not count), merged
        import cProfile
        import pstats
        import threading

        profilers = []

        def first_event(*_):
            sys.setprofile(None)
            pr = cProfile.Profile(time.thread_time)
            profilers.append(pr)
            pr.enable()

        threading.setprofile(first_event)
        t0 = time.perf_counter()
        process_qso(spectra, z_qsos, params, FixedPrior(), dla, sub, learned, batch_size=batch_size)
        out["profiled_run_seconds"] = time.perf_counter() - t0
        threading.setprofile(None)
        stats = None
        for pr in profilers:
            pr.create_stats()
            if stats is None:
                stats = pstats.Stats(pr)
            else:
                stats.add(pr)
        stats.sort_stats("tottime").print_stats(45)
    if os.environ.get("DLAGP_PROFILE"):
        import cProfile
        import pstats

        pr = cProfile.Profile()
        pr.enable()
        process_qso(spectra, z_qsos, params, FixedPrior(), dla, sub, learned, batch_size=batch_size)
        pr.disable()
        pstats.Stats(pr).sort_stats("cumtime").print_stats(25)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
