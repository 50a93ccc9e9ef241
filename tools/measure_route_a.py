#!/usr/bin/env python3
"""
INTEGRATION route A timed: the reference driver's loop body (run_bayes_select.py:141-230) with only the imports swapped --
per spectrum three GP objects, `set_data` on each, `BayesModelSelect.model_selection`, `maximum_a_posteriori`.
Synthetic DR12Q-sized spectra (tools/measure_catalogue.py), S = 10 000, max_dlas = 4.  One JSON line: spectra/s and,
with DLAGP_PROFILE=1, the cProfile table.   usage: tools/measure_route_a.py [spectra] [S]
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402

from gpy_dla_detection_amd.bayesian_model_selection import BayesModelSelect  # noqa: E402
from gpy_dla_detection_amd.dla_gp import DLAGP  # noqa: E402
from gpy_dla_detection_amd.null_gp import NullGP  # noqa: E402
from gpy_dla_detection_amd.samples import DLASamples, SubDLASamples, halton  # noqa: E402
from gpy_dla_detection_amd.set_parameters import Parameters  # noqa: E402
from gpy_dla_detection_amd.subdla_gp import SubDLAGP  # noqa: E402
from tools.measure_catalogue import FixedPrior, make_inputs  # noqa: E402


def loop(spectra, z_qsos, param, prior, dla_samples, subdla_samples, learned, max_dlas=4):
    bayes = BayesModelSelect([0, 1, max_dlas], 2)
    argmax = []
    for spectrum, z_qso in zip(spectra, z_qsos):
        np.random.seed(0)
        wavelengths, flux, noise_variance, pixel_mask = spectrum
        rest_wavelengths = param.emitted_wavelengths(wavelengths, z_qso)
        gp = NullGP(param, prior, *learned)
        gp.set_data(rest_wavelengths, flux, noise_variance, pixel_mask, z_qso, build_model=True)
        dla_gp = DLAGP(param, prior, dla_samples, *learned, min_z_separation=3000.0, broadening=True)
        dla_gp.set_data(rest_wavelengths, flux, noise_variance, pixel_mask, z_qso, build_model=True)
        subdla_gp = SubDLAGP(param, prior, subdla_samples, *learned, min_z_separation=3000.0, broadening=True)
        subdla_gp.set_data(rest_wavelengths, flux, noise_variance, pixel_mask, z_qso, build_model=True)
        bayes.model_selection([gp, subdla_gp, dla_gp], z_qso)
        dla_gp.maximum_a_posteriori()
        argmax.append(int(np.argmax(bayes.model_posteriors)))
    return argmax


def main():
    from gpy_dla_detection_amd.sightline import set_options_from_env

    options = set_options_from_env()
    num = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    param = Parameters(num_dla_samples=S)
    u = halton(S, 3)
    dla = DLASamples(param, None, halton(S, 2), 20.0 + 3.0 * u)
    sub = SubDLASamples(param, None, halton(S, 2), 19.5 + 0.5 * u, Z_dla=1.0, Z_lls=0.25)
    learned, spectra, z_qsos = make_inputs(num)
    learned = tuple(learned)[:7]
    loop(spectra[:2], z_qsos[:2], param, FixedPrior(), dla, sub, learned)
    t0 = time.perf_counter()
    argmax = loop(spectra, z_qsos, param, FixedPrior(), dla, sub, learned)
    t = time.perf_counter() - t0
    if os.environ.get("DLAGP_PROFILE"):
        import cProfile
        import pstats

        pr = cProfile.Profile()
        pr.enable()
        loop(spectra, z_qsos, param, FixedPrior(), dla, sub, learned)
        pr.disable()
        pstats.Stats(pr).sort_stats("cumtime").print_stats(30)
    from gpy_dla_detection_amd.run_bayes_select import kept_pixel_counts

    kept = kept_pixel_counts(param, spectra, z_qsos)
    # algorithmic FLOP of the run (n k (k + 1) + 2 n k per evaluation; 1 null + S sub-DLA + 4 S DLA evaluations
    # per spectrum) -- against the wall time here, and against the kernels' own time by tools/route_a_frac.py
    flop = float(kept.sum()) * (20 * 21 + 40) * (1 + 5 * S)
    out = {"route": "A (reference driver loop, imports swapped)", "spectra": num, "S": S, "max_dlas": 4,
           "seconds": t, "spectra_per_s": num / t, "ms_per_spectrum": 1e3 * t / num,
           "kept_pixels": {"min": int(kept.min()), "median": float(np.median(kept)), "mean": float(kept.mean()), "max": int(kept.max())},
           "algorithmic_gflop_per_spectrum": flop / num / 1e9, "wall_frac_of_fp64_peak": flop / t / 78.6e12,
           "argmax_counts": np.bincount(argmax, minlength=6).tolist(), "options": options}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
