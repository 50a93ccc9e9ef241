#!/usr/bin/env python3
"""
The host's share of a one-spectrum call (INTEGRATION route A), function by function: tools/measure_route_a.py's loop with
perf_counter_ns wrappers round the calls that make up a spectrum (inclusive and own time, microseconds per spectrum; the
wrappers cost ~0.3 us per call -- cProfile's 10-15 % on this loop is what they avoid).
usage: tools/route_a_host_segments.py [spectra] [S]
"""
import functools
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402

from gpy_dla_detection_amd import _lib, bayesian_model_selection, dla_gp, null_gp, samples, set_parameters, sightline, subdla_gp  # noqa: E402
from tools import measure_route_a  # noqa: E402
from tools.measure_catalogue import FixedPrior, make_inputs  # noqa: E402

TOTAL = {}   # name -> [calls, inclusive ns, own ns]
STACK = []


def timed(name, fn):
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        STACK.append(0)
        t0 = time.perf_counter_ns()
        try:
            return fn(*args, **kwargs)
        finally:
            dt = time.perf_counter_ns() - t0
            children = STACK.pop()
            if STACK:
                STACK[-1] += dt
            rec = TOTAL.setdefault(name, [0, 0, 0])
            rec[0] += 1
            rec[1] += dt
            rec[2] += dt - children
    return wrapper


def patch(owner, attr, name=None):
    name = name or "%s.%s" % (getattr(owner, "__name__", owner), attr)
    raw = owner.__dict__[attr] if isinstance(owner, type) else getattr(owner, attr)
    if isinstance(raw, staticmethod):
        setattr(owner, attr, staticmethod(timed(name, raw.__func__)))
    elif isinstance(raw, classmethod):
        setattr(owner, attr, classmethod(timed(name, raw.__func__)))
    else:
        setattr(owner, attr, timed(name, raw))


def main():
    num = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    sightline.set_options_from_env()
    for cls, attrs in ((null_gp.NullGP, ("__init__", "set_data", "device_sightline", "log_model_evidence")),
                       (dla_gp.DLAGP, ("__init__", "maximum_a_posteriori")),
                       (dla_gp._AbsorberGP, ("_sample_redshifts",)),
                       (subdla_gp.SubDLAGP, ("__init__",)),
                       (bayesian_model_selection.BayesModelSelect, ("model_selection", "_start_evidences_in_one_call")),
                       (sightline.DeviceBatch, ("__init__", "model_evidences_start", "model_evidences_finish", "_settle")),
                       (null_gp._LearnedRegistry, ("lookup",)),
                       (null_gp._SharedSightlines, ("find_or_make",)),
                       (samples.DLASamples, ("sample_z_dlas",)),
                       (set_parameters.Parameters, ("min_z_dla", "max_z_dla", "emitted_wavelengths"))):
        for attr in attrs:
            if attr in cls.__dict__:
                patch(cls, attr)
    patch(sightline.DeviceSightline, "from_model")
    for fn in ("_prepared_data", "prepare_data", "_content"):
        patch(null_gp, fn, "null_gp." + fn)
    patch(bayesian_model_selection, "settle_rng_after_sweep", "settle_rng_after_sweep")
    np.random.random_sample = timed("np.random.random_sample", np.random.random_sample)
    np.random.get_state = timed("np.random.get_state", np.random.get_state)
    lib = _lib.load()
    for entry in ("dlagp_create_from_model", "dlagp_batch_create", "dlagp_batch_model_evidences_start",
                  "dlagp_batch_model_evidences_finish", "dlagp_batch_destroy", "dlagp_destroy", "dlagp_batch_resample_ties"):
        if hasattr(lib, entry):
            try:
                setattr(lib, entry, timed("C:" + entry, getattr(lib, entry)))
            except AttributeError:
                pass

    param = set_parameters.Parameters(num_dla_samples=S)
    u = samples.halton(S, 3)
    dla = samples.DLASamples(param, None, samples.halton(S, 2), 20.0 + 3.0 * u)
    sub = samples.SubDLASamples(param, None, samples.halton(S, 2), 19.5 + 0.5 * u, Z_dla=1.0, Z_lls=0.25)
    learned, spectra, z_qsos = make_inputs(num)
    learned = tuple(learned)[:7]
    measure_route_a.loop(spectra[:2], z_qsos[:2], param, FixedPrior(), dla, sub, learned)
    TOTAL.clear()
    t0 = time.perf_counter()
    measure_route_a.loop(spectra, z_qsos, param, FixedPrior(), dla, sub, learned)
    wall = time.perf_counter() - t0
    rows = sorted(TOTAL.items(), key=lambda kv: -kv[1][1])
    table = [{"function": name, "calls_per_spectrum": round(c / num, 2), "inclusive_us": round(inc / num / 1e3, 1),
              "own_us": round(own / num / 1e3, 1)} for name, (c, inc, own) in rows]
    accounted = sum(own for _, (c, inc, own) in rows) / num / 1e3
    print(json.dumps({"spectra": num, "S": S, "wall_us_per_spectrum": round(1e6 * wall / num, 1),
                      "own_time_accounted_us": round(accounted, 1), "functions": table}, indent=1))


if __name__ == "__main__":
    main()
