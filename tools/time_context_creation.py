#!/usr/bin/env python3
"""Time of one sightline context on an otherwise idle GPU: dlagp_create_from_model / dlagp_create / destroy (per call)."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402

from gpy_dla_detection_amd.null_gp import prepare_data  # noqa: E402
from gpy_dla_detection_amd.run_bayes_select import DeviceEngine  # noqa: E402
from gpy_dla_detection_amd.set_parameters import Parameters  # noqa: E402
from gpy_dla_detection_amd.sightline import DeviceSightline  # noqa: E402
from tools.measure_catalogue import make_inputs  # noqa: E402

num = 200
params = Parameters()
learned, spectra, z_qsos = make_inputs(num)
engine = DeviceEngine(learned, params, 0)
states = []
for (wavelengths, flux, nv, mask), z in zip(spectra, z_qsos):
    states.append(prepare_data(params, params.emitted_wavelengths(wavelengths, z), flux, nv, mask, z, normalize=True))
for rep in range(2):
    t0 = time.perf_counter()
    made = [engine.sightline(st, z, True) for st, z in zip(states, z_qsos)]
    t1 = time.perf_counter()
    for m in made:
        m.close()
    t2 = time.perf_counter()
    print("from_model: create %.3f ms  destroy %.3f ms per context (n ~ %d)" % ((t1 - t0) / num * 1e3, (t2 - t1) / num * 1e3, states[0]["x"].shape[0]))
mu, M, o2 = None, None, None
made = engine.sightline(states[0], z_qsos[0], True)
mu, M, o2 = made.read_model()
st = states[0]
keep = ~st["pixel_mask"][st["ind_unmasked"]]
t0 = time.perf_counter()
devs = [DeviceSightline(st["y"], mu, M, o2, st["v"], st["padded_wavelengths"], keep) for _ in range(num)]
t1 = time.perf_counter()
print("from host arrays: create %.3f ms per context" % ((t1 - t0) / num * 1e3))
