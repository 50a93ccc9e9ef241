#!/usr/bin/env python3
"""
Soak run on the GPU box for the multi-level sweep: random batches (1-6 sightlines of 40..3300 pixels, some with mixed
flavours), S from 1 to 4000, max_dlas 2-4, random separation cuts.  Three runs of dlagp_batch_sweep per trial:
  profile cache with shared rows (production)  ==  profile cache with 16 row reads per wave     (bit for bit)
  profile cache                                ~=  every level recomputed from the Voigt profiles (base indices equal,
                                                   log-likelihoods / evidences to 1e-10)
and dlagp_batch_model_evidences against its three calls (bit for bit).
usage: python3 tools/soak_sweeps.py <seed> <trials>
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpy_dla_detection_amd import parallel, sightline as sl  # noqa: E402
from gpy_dla_detection_amd.synthetic import make_qmc_samples, make_sightline  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 100
worst, checked, shared_waves, waves = 0.0, 0, 0, 0
worst_pair, pairs = 0.0, 0
for trial in range(trials):
    nb = int(rng.choice([1, 1, 2, 3, 6]))
    S = int(rng.choice([1, 7, 16, 17, 100, 625, 1000, 2048, 4000]))
    if nb > 2 and S > 1000:
        S = 1000
    max_dlas = int(rng.choice([2, 3, 4, 4, 6]))
    sights = [make_sightline(int(rng.integers(0, 1000)), n_unmasked=int(rng.integers(40, 3300)),
                             n_masked=int(rng.integers(0, 30))) for _ in range(nb)]
    mixed = nb > 1 and rng.random() < 0.4
    devs = []
    for s in sights:
        flav = dict(num_lines=3, broadening=True, lls_break=False)
        if mixed:
            lls = bool(rng.random() < 0.25)
            flav = dict(num_lines=3 if lls else int(rng.choice([3, 3, 1, 5])), broadening=bool(rng.random() < 0.7), lls_break=lls)
        grid = s.padded_wavelengths if flav["broadening"] else s.padded_wavelengths[3:-3]
        devs.append(sl.DeviceSightline(s.y, s.this_mu, s.this_M, s.this_omega2, s.v, grid, s.keep, **flav))
    batch = sl.DeviceBatch(devs)
    zs = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    z_sub = np.stack([make_qmc_samples(s, S, sub_dla=True)[0] for s in sights])
    nhi, nhi_sub = make_qmc_samples(sights[0], S)[2], make_qmc_samples(sights[0], S, sub_dla=True)[2]
    sep = float(rng.choice([0.0, 0.003, 0.01]))
    uniforms = parallel.resampling_uniforms(nb, max_dlas, S, seed=int(rng.integers(0, 1 << 30)))
    runs = {}
    for name, cache, shared in (("shared", 1, 1), ("general", 1, 0), ("recomputed", 0, 1)):
        sl.set_option("profile_cache", cache)
        sl.set_option("shared_rows", shared)
        runs[name] = batch.sweep(zs, nhi, max_dlas, sep, uniforms)
    sl.set_option("profile_cache", 1)
    a, b, c = runs["shared"], runs["general"], runs["recomputed"]
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=x.dtype.kind == "f"), ("shared rows differ", trial)
    assert np.array_equal(a[2], c[2]), ("base indices differ from the recomputed sweep", trial)
    assert np.array_equal(np.isnan(a[1]), np.isnan(c[1]))
    ok = np.isfinite(c[1])
    if ok.any():
        rel = np.max(np.abs(a[1][ok] - c[1][ok]) / np.maximum(1.0, np.abs(c[1][ok])))
        worst = max(worst, float(rel))
        assert rel < 1e-10, (trial, rel)
    checked += int(ok.sum())
    sl.set_option("pair_models", 0)  # the three launches kept apart: bit for bit
    nulls, sub, dla = batch.model_evidences(z_sub, nhi_sub, zs, nhi, max_dlas, sep, uniforms)
    assert np.array_equal(nulls, batch.nulls())
    want_sub = batch.sweep(z_sub, nhi_sub, 1, sep)
    for x, y in zip(sub[:2], want_sub[:2]):
        assert np.array_equal(x, y, equal_nan=True), ("model_evidences: sub-DLA model", trial)
    for x, y in zip(dla, a):
        assert np.array_equal(x, y, equal_nan=x.dtype.kind == "f"), ("model_evidences: DLA model", trial)
    # round 5: the pair launch (sub-DLA model + level 1 of the DLA model, one optical depth per QMC point) and the wave's own
    # finalisation.  Automatic plans may cut the launch of 2 S rows elsewhere than the launches of S rows: agreement to
    # rounding, identical base indices and NaN pattern; with forced equal cuts (h pieces per block) everything is bit for bit,
    # pair launch or not, finished by the waves themselves (h = 1) or by finalize_rank20, fused or not.
    sl.set_option("pair_models", 1)
    p_nulls, p_sub, p_dla = batch.model_evidences(z_sub, nhi_sub, zs, nhi, max_dlas, sep, uniforms)
    pairs += sl.get_option("last_sweep_paired")
    assert np.array_equal(p_nulls, nulls)
    for got, want in ((p_sub, sub), (p_dla, dla)):
        assert np.array_equal(got[2], want[2]), ("pair launch: base indices", trial)
        for x, y in zip(got[:2], want[:2]):
            assert np.array_equal(np.isnan(x), np.isnan(y)), ("pair launch: NaN pattern", trial)
            fin = np.isfinite(y)
            if fin.any():
                rel = np.max(np.abs(x[fin] - y[fin]) / np.maximum(1.0, np.abs(y[fin])))
                worst_pair = max(worst_pair, float(rel))
                assert rel < 1e-10, ("pair launch", trial, rel)
    h = int(rng.choice([1, 1, 2, 3]))
    sl.set_option("pixel_splits", h)
    forms = []
    for pair, fuse in ((1, 1), (1, 0), (0, 1), (0, 0)):
        sl.set_option("pair_models", pair)
        sl.set_option("fuse_finalize", fuse)
        forms.append(batch.model_evidences(z_sub, nhi_sub, zs, nhi, max_dlas, sep, uniforms))
    sl.set_option("pixel_splits", 0)
    sl.set_option("pair_models", 1)
    sl.set_option("fuse_finalize", 1)
    for other in forms[1:]:
        assert np.array_equal(other[0], forms[0][0])
        for model in (1, 2):
            for x, y in zip(other[model], forms[0][model]):
                assert np.array_equal(x, y, equal_nan=x.dtype.kind == "f"), ("forced cuts: pair / fused forms differ", trial, h)
    if S >= 16:
        blocks = a[2][:, :, : S // 16 * 16].reshape(nb, max_dlas - 1, -1, 16)
        agree = (blocks == blocks[..., :1]).all(axis=-1)
        shared_waves += int(agree.sum())
        waves += agree.size
    batch.close()
    for d in devs:
        d.close()
    if trial % 20 == 19:
        print("trial", trial + 1, "values", checked, "worst", worst, flush=True)
print("seed", sys.argv[1] if len(sys.argv) > 1 else 0, "trials", trials, "values", checked, "worst relative difference", worst,
      "waves with a shared row", shared_waves, "of", waves, "| pair launches", pairs, "worst pair-vs-separate difference", worst_pair)
