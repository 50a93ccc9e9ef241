#!/usr/bin/env python3
"""
Soak run on the GPU box for the level finish and the exactness of the resampling (round 6).  Random batches (1-4 sightlines of
60..1300 pixels), S from 1 to 12 000, max_dlas 2-4, ordinary uniforms.  Per trial:
  level finish spread over many workgroups (random number of search groups)  ==  one workgroup per sightline   (bit for bit)
  EVERY level of every sightline -- not only the ones the library flagged -- against NumPy's own
  cumsum / searchsorted on the level's likelihoods as returned (`sightline.exact_choice`): the returned base indices must be
  NumPy's.  An unflagged level whose indices differed would mean DLAGP_TIE_MARGIN is too small; the run counts flagged
  levels, levels the host changed, and (must stay 0) unflagged differences.
usage: python3 tools/soak_levels.py <seed> <trials>
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpy_dla_detection_amd import sightline as sl  # noqa: E402
from gpy_dla_detection_amd.synthetic import make_qmc_samples, make_sightline  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 100
levels = flagged = changed = unflagged_differences = form_differences = searches = 0
for trial in range(trials):
    nb = int(rng.choice([1, 1, 2, 4]))
    S = int(rng.choice([1, 2, 63, 64, 65, 1000, 1024, 1025, 3000, 10000, 12000]))
    if nb > 2 and S > 3000:
        S = 3000
    max_dlas = int(rng.choice([2, 3, 4, 4]))
    sights = [make_sightline(int(rng.integers(0, 1000)), n_unmasked=int(rng.integers(60, 1300)),
                             n_masked=int(rng.integers(0, 20))) for _ in range(nb)]
    devs = [sl.DeviceSightline.from_state(s) for s in sights]
    batch = sl.DeviceBatch(devs)
    z = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    nhi = make_qmc_samples(sights[0], S)[2]
    min_sep = float(rng.choice([0.0, 0.01, 0.05]))
    uniforms = rng.random((nb, max_dlas - 1, S))
    sl.set_option("level_spread", 0)
    ev0, ll0, base0 = batch.sweep(z, nhi, max_dlas, min_sep, uniforms)
    ties0 = batch.last_ties.copy()
    sl.set_option("level_spread", 1)
    sl.set_option("level_search_groups", int(rng.choice([1, 3, 16, 64])))
    ev, ll, base = batch.sweep(z, nhi, max_dlas, min_sep, uniforms)
    same = (np.array_equal(ev, ev0, equal_nan=True) and np.array_equal(ll, ll0, equal_nan=True) and np.array_equal(base, base0)
            and np.array_equal(batch.last_ties, ties0))
    form_differences += 0 if same else 1
    for b in range(nb):
        for level in range(max_dlas - 1):
            if np.isnan(ev[b, level]):
                break
            levels += 1
            searches += S
            exact = sl.exact_choice(ll[b, :, level], uniforms[b, level])
            if batch.last_ties[b, level]:
                flagged += 1
            if not np.array_equal(exact, base[b, level]):
                unflagged_differences += 1  # after settle_near_ties nothing may differ, flagged or not
    changed += batch.last_changed if batch.last_ties.any() else 0
    batch.close()
    for d in devs:
        d.close()
report = {"trials": trials, "levels_checked": levels, "searches": searches, "levels_flagged": flagged,
          "levels_changed_by_the_host": changed, "levels_whose_returned_indices_are_not_numpys": unflagged_differences,
          "trials_where_the_two_forms_differ": form_differences}
print(json.dumps(report))
sys.exit(0 if unflagged_differences == 0 and form_differences == 0 else 1)
