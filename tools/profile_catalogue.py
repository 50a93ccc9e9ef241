#!/usr/bin/env python3
"""
Where a catalogue run's wall time goes (one GPU): the main thread's phases of run_bayes_select.process_qso timed by
wrapping the engine's methods -- waiting for the prepared batch, null evidences, the two sweeps -- plus the worker
threads' prepare / finish totals.  usage: tools/profile_catalogue.py [spectra] [S] [batch]
"""
import json
import os
import sys
import threading
import time
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402

from gpy_dla_detection_amd import run_bayes_select as rbs  # noqa: E402
from gpy_dla_detection_amd.samples import DLASamples, SubDLASamples, halton  # noqa: E402
from gpy_dla_detection_amd.set_parameters import Parameters  # noqa: E402
from tools.measure_catalogue import FixedPrior, make_inputs  # noqa: E402

totals = defaultdict(float)
lock = threading.Lock()
events = []  # (name, thread, start, end) of the engine-level calls, for a timeline (DLAGP_TIMELINE=<file>)


def timed(name, fn):
    def wrapper(*args, **kwargs):
        t0 = time.perf_counter()
        try:
            return fn(*args, **kwargs)
        finally:
            t1 = time.perf_counter()
            with lock:
                totals[name] += t1 - t0
                if not name.startswith(("C:", "py:")):
                    events.append((name, threading.get_ident(), t0, t1))
    return wrapper


def main():
    num = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    batch_size = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    params = Parameters(num_dla_samples=S)
    u = halton(S, 3)
    dla = DLASamples(params, None, halton(S, 2), 20.0 + 3.0 * u)
    sub = SubDLASamples(params, None, halton(S, 2), 19.5 + 0.5 * u, Z_dla=1.0, Z_lls=0.25)
    learned, spectra, z_qsos = make_inputs(num)
    rbs.process_qso(spectra[:batch_size], z_qsos[:batch_size], params, FixedPrior(), dla, sub, learned, batch_size=batch_size)
    for name in ("sightline", "batch", "nulls", "sweep", "release"):
        setattr(rbs.DeviceEngine, name, timed(name, getattr(rbs.DeviceEngine, name)))
    from gpy_dla_detection_amd import _lib, parallel, sightline

    lib = _lib.load()
    for cname in ("dlagp_batch_sweep", "dlagp_batch_null", "dlagp_create_from_model", "dlagp_batch_create",
                  "dlagp_batch_destroy", "dlagp_destroy"):
        setattr(lib, cname, timed("C:" + cname, getattr(lib, cname)))
    sightline.DeviceBatch.sweep = timed("py:DeviceBatch.sweep", sightline.DeviceBatch.sweep)
    parallel.resampling_uniforms = timed("py:resampling_uniforms", parallel.resampling_uniforms)
    from concurrent.futures import Future

    Future.result = timed("main:future.result (waiting for a worker)", Future.result)
    rbs.prepare_data = timed("prepare_data", rbs.prepare_data)
    rbs.map_traceback = timed("map_traceback", rbs.map_traceback)
    rbs.occupation_log_priors = timed("log_priors", rbs.occupation_log_priors)
    events.clear()
    t0 = time.perf_counter()
    rbs.process_qso(spectra, z_qsos, params, FixedPrior(), dla, sub, learned, batch_size=batch_size)
    wall = time.perf_counter() - t0
    if os.environ.get("DLAGP_TIMELINE"):
        threads = {t: i for i, t in enumerate(sorted({e[1] for e in events}))}
        with open(os.environ["DLAGP_TIMELINE"], "w") as f:
            json.dump([(n, threads[t], round((a - t0) * 1e3, 3), round((b - t0) * 1e3, 3)) for n, t, a, b in events], f)
    out = {"spectra": num, "S": S, "batch_size": batch_size, "wall_s": wall, "spectra_per_s": num / wall,
           "seconds_inside": {k: round(v, 4) for k, v in sorted(totals.items())}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
