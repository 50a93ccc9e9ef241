#!/usr/bin/env python3
"""
CPU baseline variants of SURVEY.md §8(d) on this host, config 2 inputs (n = 3000, k = 20, one absorber):
  (i)   1 process, BLAS threads = 1        (ii) 1 process, BLAS threads = default
  (iii) one process per core (bench.py's cpu_baseline; printed by bench.py itself)
The oracle (oracle/ref_like.py) is the evaluator: the reference's per-sample loop in NumPy/SciPy.
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from threadpoolctl import threadpool_limits

from gpy_dla_detection_amd.synthetic import make_qmc_samples, make_sightline
from oracle import ref_like


def rate(st, z, nhi, seconds=6.0):
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        ref_like.sample_log_likelihood_k_dlas(st, [z[n % len(z)]], [nhi[n % len(z)]])
        n += 1
    return n / (time.perf_counter() - t0)


def main():
    sight = make_sightline(0)
    z, _, nhi = make_qmc_samples(sight, 10000)
    st = ref_like.SpectrumState(sight.y, sight.v, sight.this_mu, sight.this_M, sight.this_omega2,
                                sight.padded_wavelengths, sight.unmasked_wavelengths, sight.keep)
    out = {"host_cpus": os.cpu_count(), "numpy": np.__version__}
    with threadpool_limits(limits=1):
        out["one_process_blas1_evals_per_s"] = rate(st, z, nhi)
    out["one_process_blas_default_evals_per_s"] = rate(st, z, nhi)
    try:
        out["loadavg"] = os.getloadavg()[0]
    except OSError:
        pass
    print(json.dumps(out))


if __name__ == "__main__":
    main()
