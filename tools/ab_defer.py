import os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
from gpy_dla_detection_amd import run_bayes_select as rbs
from gpy_dla_detection_amd.samples import DLASamples, SubDLASamples, halton
from gpy_dla_detection_amd.set_parameters import Parameters
from tools.measure_catalogue import FixedPrior, make_inputs
S = 10000
params = Parameters(num_dla_samples=S)
u = halton(S, 3)
dla = DLASamples(params, None, halton(S, 2), 20.0 + 3.0 * u)
sub = SubDLASamples(params, None, halton(S, 2), 19.5 + 0.5 * u, Z_dla=1.0, Z_lls=0.25)
learned, spectra, z_qsos = make_inputs(1024)
orig = rbs.DeviceEngine.__init__
for rep in range(2):
    for defer in (True, False):
        def init(self, *a, _d=defer, **k):
            orig(self, *a, **k); self.defer_sync = _d
        rbs.DeviceEngine.__init__ = init
        for num, bs in ((512, 32), (1024, 64)):
            rbs.process_qso(spectra[:bs], z_qsos[:bs], params, FixedPrior(), dla, sub, learned, batch_size=bs)
            t0 = time.perf_counter()
            rbs.process_qso(spectra[:num], z_qsos[:num], params, FixedPrior(), dla, sub, learned, batch_size=bs)
            t = time.perf_counter() - t0
            print("defer=%s  %d spectra, batches of %d: %.1f spectra/s" % (defer, num, bs, num / t), flush=True)
