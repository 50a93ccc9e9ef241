#!/usr/bin/env python3
"""cProfile of run_bayes_select.process_qso with prefetch=False (everything in one thread): where the Python time of a
catalogue run goes, by function.  usage: tools/profile_catalogue_serial.py [spectra] [S] [batch]"""
import cProfile
import os
import pstats
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from gpy_dla_detection_amd import run_bayes_select as rbs  # noqa: E402
from gpy_dla_detection_amd.samples import DLASamples, SubDLASamples, halton  # noqa: E402
from gpy_dla_detection_amd.set_parameters import Parameters  # noqa: E402
from tools.measure_catalogue import FixedPrior, make_inputs  # noqa: E402

num = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
batch_size = int(sys.argv[3]) if len(sys.argv) > 3 else 32
params = Parameters(num_dla_samples=S)
u = halton(S, 3)
dla = DLASamples(params, None, halton(S, 2), 20.0 + 3.0 * u)
sub = SubDLASamples(params, None, halton(S, 2), 19.5 + 0.5 * u, Z_dla=1.0, Z_lls=0.25)
learned, spectra, z_qsos = make_inputs(num)
rbs.process_qso(spectra[:batch_size], z_qsos[:batch_size], params, FixedPrior(), dla, sub, learned, batch_size=batch_size, prefetch=False)
pr = cProfile.Profile()
pr.enable()
rbs.process_qso(spectra, z_qsos, params, FixedPrior(), dla, sub, learned, batch_size=batch_size, prefetch=False)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
