#!/usr/bin/env python3
"""
Soak run on the GPU box: random batches (1-9 sightlines of 9..3300 pixels, random and run-shaped masks, 1-4 absorbers per
sample, 1/3/5/31 lines, with and without broadening / the Lyman-limit term, S from 1 to 10 000, forced and automatic work
decompositions; since round 3 also batches whose members differ in flavour), MFMA kernel (path 2) against the generic
kernel (path 1) on the same inputs.
usage: python3 tools/soak_paths.py <seed> <trials>      (round 2: seeds 1-3 x 300 trials, 3.0 M values, worst 4.4e-11)
"""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gpy_dla_detection_amd import sightline as sl
from gpy_dla_detection_amd.synthetic import make_qmc_samples, make_sightline
from tests.golden_inputs import gapped_sightline
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
checked = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 150):
    nb = int(rng.choice([1, 1, 1, 2, 3, 5, 9]))
    S = int(rng.choice([1, 15, 16, 17, 100, 625, 1000, 2048, 4000, 10000]))
    if nb > 2 and S > 2048: S = 2048
    kd = int(rng.choice([1, 1, 2, 3, 4]))
    lls = bool(rng.random() < 0.2)
    broad = bool(rng.random() < 0.85)
    lines = int(rng.choice([3, 3, 3, 1, 5, 31]))
    if lls: lines = 3
    sights = []
    rank = int(rng.choice([20, 20, 20, 12, 24, 33, 64]))  # one rank per batch; above 20: column groups of the MFMA kernel
    for b in range(nb):
        n_u = int(rng.integers(9, 3300))
        if rng.random() < 0.3 and n_u > 400:
            lo = int(rng.integers(5, n_u // 2)); run = int(rng.integers(30, n_u // 3))
            s = gapped_sightline(index=int(rng.integers(0, 1000)), n_unmasked=n_u, gaps=((lo, lo + run),), comb=None, k=rank)
        else:
            s = make_sightline(int(rng.integers(0, 1000)), n_unmasked=n_u, n_masked=int(rng.integers(0, max(1, n_u // 10))), k=rank)
        sights.append(s)
    mixed = nb > 1 and rng.random() < 0.4  # every member its own flavour: one MFMA launch group per flavour
    devs = []
    for s in sights:
        if mixed:
            m_lls = bool(rng.random() < 0.2)
            flav = dict(num_lines=3 if m_lls else int(rng.choice([3, 3, 1, 5, 31])), broadening=bool(rng.random() < 0.7), lls_break=m_lls)
        else:
            flav = dict(num_lines=lines, broadening=broad, lls_break=lls)
        devs.append(sl.DeviceSightline.from_state(s, **flav))
    zz = np.stack([np.stack([np.roll(make_qmc_samples(s, S)[0], 3 * j) for j in range(kd)], axis=1) for s in sights])
    nn = np.stack([np.stack([np.roll(make_qmc_samples(s, S)[2], 5 * j) for j in range(kd)], axis=1) for s in sights])
    splits = int(rng.choice([0, 0, 0, 1, 2, 5]))
    sl.set_option("pixel_splits", splits)
    if nb == 1:
        ev = lambda: devs[0].eval(zz[0], nn[0])[None, :]
    else:
        batch = sl.DeviceBatch(devs)
        ev = lambda: batch.eval(zz, nn)
    sl.set_option("path", 2)
    got = ev()
    assert sl.get_option("last_launch_path") == 2
    sl.set_option("path", 1)
    want = ev()
    sl.set_option("path", 0)
    assert np.array_equal(np.isnan(got), np.isnan(want)), (trial, nb, S, kd)
    fin = np.isfinite(want)
    rel = np.max(np.abs(got[fin] - want[fin]) / np.maximum(np.abs(want[fin]), 1.0)) if fin.any() else 0.0
    worst = max(worst, rel)
    checked += int(fin.sum())
    assert rel < 5e-11, (trial, nb, S, kd, lls, broad, lines, splits, rel)
    if nb > 1: batch.close()
    for d in devs: d.close()
print("soak ok: %d trials, %d values, worst relative difference MFMA vs generic kernel %.2e" % (trial + 1, checked, worst))
