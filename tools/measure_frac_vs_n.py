#!/usr/bin/env python3
"""
Efficiency of the accumulate kernels against the sightline length (round 5).

Real DR12Q sightlines keep n <= 1250 pixels of the modelled window (reference set_parameters.py:33-34,58,
null_gp.py:139-152; median ~685 for a DR12Q-like redshift spread), the bench's synthetic ones 3000.  For a batch of
`--sightlines` (default 32) synthetic sightlines of n_u = `--pixels` unmasked pixels, S = 10 000 QMC points, ONE stream:

  sub_dla      one-absorber sweep without the profile cache (the sub-DLA model / DLA(1) of configs 3 and 4)
  level1       level 1 of a cached multi-level sweep (also stores the profiles)
  cached       one deeper level of that sweep

each as {ms per launch (HIP events on the launch's dispatch packet), evals per launch, frac of 78.6 TFLOP/s from
F = n k (k + 1) + 2 n k per eval}, plus the wall time of a whole step (finalisation, level_finish included), the launch
plan's shape and -- with --stamps -- the per-phase cycle stamps of the one-absorber kernel.

    python3 tools/measure_frac_vs_n.py --pixels 685 [--sightlines 32] [--steps 5] [--stamps] [--option name=value ...]
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/measure_frac_vs_n.py --pixels 685

Prints one JSON object.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gpy_dla_detection_amd import parallel  # noqa: E402
from gpy_dla_detection_amd import sightline as sl  # noqa: E402
from gpy_dla_detection_amd.synthetic import make_qmc_samples, make_sightline  # noqa: E402

PEAK = 78.6
MIN_Z_SEPARATION = 3000.0e3 / 299792458.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pixels", type=int, default=685, help="unmasked pixels n_u of every sightline (2 %% of them masked)")
    ap.add_argument("--spread", type=float, default=0.0,
                    help="> 0: n_u drawn uniformly from [pixels (1 - spread), pixels (1 + spread)] per sightline")
    ap.add_argument("--sightlines", type=int, default=32)
    ap.add_argument("--samples", type=int, default=10000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--stamps", action="store_true")
    ap.add_argument("--option", action="append", default=[])
    ap.add_argument("--no-cached", action="store_true", help="skip the multi-level sweeps")
    ap.add_argument("--catalogue", choices=["uniform", "dr12q"], default=None,
                    help="the batch is the first `--sightlines` spectra of tools/measure_catalogue.py's synthetic catalogue (raw "
                         "spectra -> prepare_data -> contexts from the learned model, sample redshifts from the spectrum's own "
                         "range) instead of synthetic sightlines of a given length")
    ap.add_argument("--skip", type=int, default=0, help="--catalogue: first spectrum of the batch")
    ap.add_argument("--noise", choices=["none", "contexts", "transfers"], default="none",
                    help="what a side thread does beside the timed sweeps: `contexts` creates and destroys sightline contexts "
                         "from raw spectra the way the catalogue driver's preparing thread does (copies + set-up kernels on "
                         "its own stream), `transfers` moves 27 MB to and from pinned host memory per 12 ms on its own stream")
    ap.add_argument("--gap-ms", type=float, default=0.0,
                    help="> 0: the device is left idle for this long after every timed step (does the clock fall in a pipeline's pauses?)")
    ap.add_argument("--flush-mb", type=int, default=0,
                    help="> 0: a buffer of this size is overwritten in front of every timed step, so that a step finds its "
                         "operands in HBM and not in the last-level cache (a catalogue's batches are new data every time)")
    args = ap.parse_args()

    nb, S = args.sightlines, args.samples
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    sights = []
    cat_engine = None
    if args.catalogue:
        from gpy_dla_detection_amd.null_gp import prepare_data
        from gpy_dla_detection_amd.run_bayes_select import DeviceEngine
        from gpy_dla_detection_amd.samples import DLASamples, halton
        from gpy_dla_detection_amd.set_parameters import Parameters
        from tools.measure_catalogue import make_inputs

        params = Parameters(num_dla_samples=S)
        learned, spectra, z_qsos = make_inputs(args.skip + nb, mix=args.catalogue)
        cat_engine = DeviceEngine(learned, params)
        holder = DLASamples(params, None, halton(S, 2), 20.0 + 3.0 * halton(S, 3))
        devs, z_rows, kept_n = [], [], []
        for (w, f, v, m), zq in list(zip(spectra, z_qsos))[args.skip:]:
            st = prepare_data(params, params.emitted_wavelengths(w, zq), f, v, m, zq, normalize=True)
            devs.append(cat_engine.sightline(st, zq, True))
            z_rows.append(holder.sample_z_dlas(st["this_wavelengths"], zq))
            kept_n.append(len(st["x"]))
        n_mean = float(np.mean(kept_n))
        flops = float(np.mean([n * 20 * 21 + 2 * n * 20 for n in kept_n]))
    else:
        for i in range(nb):
            n_u = args.pixels if args.spread <= 0 else int(round(args.pixels * (1 + args.spread * (2 * rng.random() - 1))))
            sights.append(make_sightline(i, n_unmasked=n_u, n_masked=max(1, round(n_u * 61 / 3061)), k=20))
        n_mean = float(np.mean([s.n for s in sights]))
        flops = float(np.mean([s.n * 20 * 21 + 2 * s.n * 20 for s in sights]))
        devs = [sl.DeviceSightline.from_state(s, device=0) for s in sights]
    batch = sl.DeviceBatch(devs)
    for item in args.option:
        name, value = item.split("=")
        sl.set_option(name, int(value))
    f64 = dict(dtype=torch.float64, device=dev)
    if args.catalogue:
        z = torch.from_numpy(np.stack(z_rows)).to(dev)
        nhi_dla = torch.from_numpy(10.0 ** (20.0 + 3.0 * halton(S, 3))).to(dev)
        nhi_sub = torch.from_numpy(10.0 ** (19.5 + 0.5 * halton(S, 3))).to(dev)
    else:
        z = torch.from_numpy(np.stack([make_qmc_samples(s, S)[0] for s in sights])).to(dev)
        nhi_dla = torch.from_numpy(make_qmc_samples(sights[0], S)[2]).to(dev)
        nhi_sub = torch.from_numpy(make_qmc_samples(sights[0], S, sub_dla=True)[2]).to(dev)
    uniforms = torch.from_numpy(parallel.resampling_uniforms(nb, 4, S, seed=0)).to(dev)
    evid = torch.zeros(nb * 4, **f64)
    sample_ll = torch.empty((nb, 4, S), **f64)
    base = torch.zeros((nb, 3, S), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    flush = torch.empty(args.flush_mb << 20, dtype=torch.uint8, device=dev) if args.flush_mb > 0 else None

    noise_stop, noise_thread, noise_count = None, None, [0]
    if args.noise != "none":
        import threading

        noise_stop = threading.Event()
        if args.noise == "contexts":
            from gpy_dla_detection_amd.null_gp import prepare_data
            from gpy_dla_detection_amd.run_bayes_select import DeviceEngine
            from gpy_dla_detection_amd.set_parameters import Parameters
            from tools.measure_catalogue import make_inputs

            params = Parameters(num_dla_samples=S)
            learned, spectra, z_qsos = make_inputs(64, mix="dr12q")
            engine = DeviceEngine(learned, params)
            states = []
            for (w, f, v, m), zq in zip(spectra, z_qsos):
                states.append((prepare_data(params, params.emitted_wavelengths(w, zq), f, v, m, zq, normalize=True), zq))

            def noise():
                torch.cuda.set_device(0)
                while not noise_stop.is_set():
                    made = [engine.sightline(st, zq, True) for st, zq in states[:32]]
                    engine.batch(made).close()  # waits for the set-up work, as the driver's batch creation does
                    for c in made:
                        c.close()
                    noise_count[0] += len(made)
        else:
            host = torch.empty(27 << 20, dtype=torch.uint8).pin_memory()
            there = torch.empty(27 << 20, dtype=torch.uint8, device=dev)
            side = torch.cuda.Stream()

            def noise():
                torch.cuda.set_device(0)
                while not noise_stop.is_set():
                    with torch.cuda.stream(side):
                        there.copy_(host, non_blocking=True)
                        host.copy_(there, non_blocking=True)
                    side.synchronize()
                    noise_count[0] += 1
                    time.sleep(0.010)
        noise_thread = threading.Thread(target=noise, daemon=True)
        noise_thread.start()

    def sweep(max_dlas, nhi):
        batch.sweep_device(S, max_dlas, z.data_ptr(), nhi.data_ptr(), 0, MIN_Z_SEPARATION,
                           uniforms.data_ptr() if max_dlas > 1 else 0, evid.data_ptr(), sample_ll.data_ptr(),
                           base.data_ptr() if max_dlas > 1 else 0, stream)

    def timed(fn, steps):
        """(wall ms per step, accumulate ms per step, accumulate launches per step)"""
        fn()
        torch.cuda.synchronize()
        t_ramp = time.perf_counter()
        while (time.perf_counter() - t_ramp) < 0.15:  # clock ramp
            fn()
            torch.cuda.synchronize()
        sl.set_option("timing", 1)
        sl.timing_read(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            if flush is not None:
                flush.zero_()  # same stream: runs between the steps (its time is in `wall`, not in the accumulate time)
            fn()
            if args.gap_ms > 0:
                torch.cuda.synchronize()
                time.sleep(args.gap_ms * 1e-3)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3 / steps
        sl.set_option("timing", 0)
        ms, launches = sl.timing_read(reset=True)
        return wall, ms / steps, launches // steps

    def entry(ms, evals):
        tf = evals * flops / (ms * 1e-3) / 1e12
        return {"ms_per_launch": ms, "evals_per_launch": evals, "tflops": tf, "frac": tf / PEAK}

    out = {"pixels_unmasked": args.pixels, "spread": args.spread, "n_kept_mean": n_mean, "sightlines": nb, "samples": S,
           "flop_per_eval": flops, "options": args.option, "flush_mb": args.flush_mb, "catalogue": args.catalogue, "gap_ms": args.gap_ms}
    evals = float(nb) * S
    wall1, acc1, _ = timed(lambda: sweep(1, nhi_sub), args.steps)
    out["sub_dla"] = entry(acc1, evals)
    out["sub_dla"]["step_wall_ms"] = wall1
    out["sub_dla"]["other_kernels_share"] = 1.0 - acc1 / wall1
    out["sub_dla"]["step_frac"] = evals * flops / (wall1 * 1e-3) / 1e12 / PEAK  # finalisation and level_finish included
    out["sub_dla"]["fused_finalize"] = sl.get_option("last_launch_fused")
    out["plan"] = {"work_items": sl.get_option("last_launch_items"), "pieces_per_block": sl.get_option("last_launch_parts"),
                   "slots_mean": float(np.mean([-(-s.n // 8) * 8 for s in sights])) if sights else None}
    # the sub-DLA sweep and DLA(1) in one call: one launch for both when the batch's shape allows (pair launch)
    sub_ev, sub_ll = torch.zeros(nb, **f64), torch.empty((nb, 1, S), **f64)

    def two_models(max_dlas):
        batch.two_model_sweep_device(S, max_dlas, z.data_ptr(), nhi_sub.data_ptr(), 0, nhi_dla.data_ptr(), 0, MIN_Z_SEPARATION,
                                     uniforms.data_ptr() if max_dlas > 1 else 0, sub_ev.data_ptr(), sub_ll.data_ptr(),
                                     evid.data_ptr(), sample_ll.data_ptr(), base.data_ptr() if max_dlas > 1 else 0, stream)

    wall_p, acc_p, launches_p = timed(lambda: two_models(1), args.steps)
    out["sub_dla_plus_dla1"] = entry(acc_p, 2 * evals)
    out["sub_dla_plus_dla1"].update(step_wall_ms=wall_p, launches=launches_p, paired=sl.get_option("last_sweep_paired"),
                                    step_frac=2 * evals * flops / (wall_p * 1e-3) / 1e12 / PEAK)
    if args.stamps:
        sl.set_option("debug", 2)
        sl.debug_read(reset=True)
        sweep(1, nhi_sub)
        torch.cuda.synchronize()
        st = sl.debug_read()
        sl.set_option("debug", 0)
        if st[5]:
            names = ["voigt", "conv", "mfma_loop", "epilogue", "passes_and_epilogue", None, "prologue"]
            out["stamps_cycles_per_wave"] = {n: st[i] / st[5] for i, n in enumerate(names) if n}
            out["stamps_cycles_per_wave"]["waves"] = st[5]
    if not args.no_cached:
        wall2, acc2, l2 = timed(lambda: sweep(2, nhi_dla), args.steps)
        wall4, acc4, l4 = timed(lambda: sweep(4, nhi_dla), args.steps)
        cached = (acc4 - acc2) / 2.0
        out["cached_level"] = entry(cached, evals)
        out["level1_with_store"] = entry(acc2 - cached, evals)
        out["sweep4"] = {"step_wall_ms": wall4, "accumulate_ms": acc4, "launches": l4,
                         "other_kernels_share": 1.0 - acc4 / wall4, "frac": 4 * evals * flops / (acc4 * 1e-3) / 1e12 / PEAK,
                         "step_frac": 4 * evals * flops / (wall4 * 1e-3) / 1e12 / PEAK,
                         "cached": sl.get_option("last_sweep_cached"), "fused_finalize": sl.get_option("last_launch_fused")}
        # what a catalogue run does per batch: null + sub-DLA + DLA(1..4) (the null launch is not part of this figure)
        wall5, acc5, l5 = timed(lambda: two_models(4), args.steps)
        out["model_selection_5_levels"] = {"step_wall_ms": wall5, "accumulate_ms": acc5, "launches": l5,
                                           "paired": sl.get_option("last_sweep_paired"),
                                           "frac": 5 * evals * flops / (acc5 * 1e-3) / 1e12 / PEAK,
                                           "step_frac": 5 * evals * flops / (wall5 * 1e-3) / 1e12 / PEAK}
    if noise_thread is not None:
        noise_stop.set()
        noise_thread.join()
        out["noise"] = {"kind": args.noise, "contexts_made_or_rounds": noise_count[0]}
    print(json.dumps(out))
    batch.close()
    for d in devs:
        d.close()


if __name__ == "__main__":
    main()
