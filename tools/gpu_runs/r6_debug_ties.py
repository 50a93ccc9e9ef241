import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gpy_dla_detection_amd import parallel, sightline as sl
from gpy_dla_detection_amd.synthetic import make_qmc_samples, make_sightline
from tests.test_gpu_parity import numpy_cdf, planted_uniforms

rng = np.random.default_rng(11)
sights = [make_sightline(80, n_unmasked=310, n_masked=6)]
batch = sl.DeviceBatch([sl.DeviceSightline.from_state(s) for s in sights])
S, max_dlas, nb = 3000, 3, 1
zs = np.stack([make_qmc_samples(s, S)[0] for s in sights]); nhi = make_qmc_samples(sights[0], S)[2]
min_sep = 3000.0e3 / 299792458.0
uniforms = parallel.resampling_uniforms(nb, max_dlas, S, seed=2)
orig = sl.settle_near_ties
def spy(ties, *a, **k):
    print("ties from device:", ties.tolist())
    return orig(ties, *a, **k)
sl.settle_near_ties = spy
ev, ll, base = batch.sweep(zs, nhi, max_dlas, min_sep, uniforms)
print("random: last_ties", batch.last_ties)
cdf = numpy_cdf(ll[0, :, 0])
uniforms[0, 0] = planted_uniforms(cdf, rng)
ev, ll, base = batch.sweep(zs, nhi, max_dlas, min_sep, uniforms)
print("planted: last_ties", batch.last_ties, "changed", batch.last_changed)
want = cdf.searchsorted(uniforms[0, 0], side="right")
print("mismatch after settle", int((want != base[0, 0]).sum()))
