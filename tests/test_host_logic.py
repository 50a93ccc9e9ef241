"""
CPU tests of the host-side logic around the device path: data preparation (set_data/get_interp,
f1 row), the occupation sweep's level logic with the legacy-RNG resampling, the batch driver and
the multi-process sharding + gather (gloo, world_size 2).  Where a log-likelihood evaluator is
needed the oracle stands in for the GPU -- as the checker of the host logic, never as a product path.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from gpy_dla_detection_amd import parallel
from gpy_dla_detection_amd.dla_gp import occupation_sweep
from gpy_dla_detection_amd.samples import halton
from gpy_dla_detection_amd.set_parameters import Parameters
from gpy_dla_detection_amd.synthetic import make_qmc_samples, make_sightline
from tests.conftest import REPO, load_golden, oracle_state
from tests.golden_inputs import checksum, raw_spectrum_and_model


class FixedPrior:
    def less_ind(self, z_qso):
        return 731.0, 9265.0


def test_set_data_and_get_interp_match_reference_golden():
    from gpy_dla_detection_amd.null_gp import NullGP

    g = load_golden("set_data.npz")
    z_qso, rest, flux, nv, mask, rest_grid, mu, M, log_omega, lc0, lt0, lb = raw_spectrum_and_model()
    params = Parameters()
    gp = NullGP(params, FixedPrior(), rest_grid, mu, M, log_omega, lc0, lt0, lb)
    gp.set_data(rest, flux, nv, mask, z_qso, normalize=True, build_model=True)
    for name in ("x", "y", "v", "this_mu", "this_M", "this_omega2", "padded_wavelengths",
                 "unmasked_wavelengths", "this_wavelengths"):
        np.testing.assert_allclose(getattr(gp, name), g[name], rtol=1e-14, atol=0, err_msg=name)
    np.testing.assert_array_equal(gp.ind_unmasked, g["ind_unmasked"])
    assert gp.this_M.flags["F_CONTIGUOUS"]  # the layout the reference hands to the sweep
    assert abs(gp.normalization_median - g["normalization_median"]) <= 1e-15 * abs(g["normalization_median"])
    assert abs(params.min_z_dla(gp.this_wavelengths, z_qso) - g["min_z_dla"]) < 1e-15
    assert abs(params.max_z_dla(gp.this_wavelengths, z_qso) - g["max_z_dla"]) < 1e-15


def test_parameters_surface():
    p = Parameters()
    assert p.k == 20 and p.num_dla_samples == 10000 and p.num_lines == 3 and p.width == 3
    assert abs(Parameters.kms_to_z(3000) - 0.01) < 1e-4  # reference tests/test_set_parameters.py
    assert abs(p.max_z_cut - 3000e3 / 299792458.0) < 1e-18
    assert Parameters(910.0, 1217.0, k=7).k == 7
    with pytest.raises(TypeError):
        Parameters(no_such_parameter=1)


def test_halton_and_shards():
    h = halton(8, 2)
    np.testing.assert_allclose(h, [0.5, 0.25, 0.75, 0.125, 0.625, 0.375, 0.875, 0.0625])
    for total in (0, 1, 7, 64, 1024):
        for world in (1, 2, 3, 8):
            edges = [parallel.shard_range(total, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


def oracle_level_evaluator(sight):
    from oracle import ref_like as R

    st = oracle_state(sight)

    def evaluate(z, nhi):
        return np.array([R.sample_log_likelihood_k_dlas(st, z[i], nhi[i]) for i in range(z.shape[0])])

    return evaluate


def test_occupation_sweep_host_logic_matches_reference_golden():
    """level logic + legacy-RNG resampling of this package, with the oracle as the evaluator"""
    g = load_golden("evidences_small.npz")
    sight = make_sightline(int(g["sight_index"]), n_unmasked=int(g["sight_n_unmasked"]),
                           n_masked=int(g["sight_n_masked"]), k=int(g["sight_k"]))
    assert checksum(sight.y, sight.this_M, sight.padded_wavelengths, sight.keep) == int(g["sight_crc"])
    S = int(g["S"])
    nhi = 10.0 ** (20.0 + 3.0 * halton(S, 3))
    np.random.seed(0)
    ev, sll, base = occupation_sweep(oracle_level_evaluator(sight), g["dla_sample_z"], nhi, 4,
                                     float(g["dla_min_z_separation"]), S)
    np.testing.assert_array_equal(base, g["dla_base_inds"])
    np.testing.assert_allclose(ev, g["dla_evidences"], rtol=1e-12)
    np.testing.assert_allclose(sll, g["dla_sample_ll"], rtol=1e-12, equal_nan=True)


def test_sweep_batch_equals_sightline_by_sightline():
    S, max_dlas = 48, 3
    sights = [make_sightline(i, n_unmasked=160 + 8 * i, n_masked=3) for i in range(3)]
    evals = [oracle_level_evaluator(s) for s in sights]
    z = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    nhi = make_qmc_samples(sights[0], S)[2]

    def batch_eval(zb, nb):
        return np.stack([evals[b](zb[b], nb[b]) for b in range(len(sights))])

    np.random.seed(123)
    marker = np.random.random()  # the caller's RNG stream must come back untouched
    np.random.seed(123)
    ev, sll, base = parallel.sweep_batch(batch_eval, z, nhi, max_dlas, 0.01, seed=0)
    assert np.random.random() == marker
    for b, s in enumerate(sights):
        np.random.seed(0)  # the reference driver re-seeds per spectrum (run_bayes_select.py:144)
        ev1, sll1, base1 = occupation_sweep(evals[b], z[b], nhi, max_dlas, 0.01, S)
        np.testing.assert_array_equal(base[b], base1)
        np.testing.assert_array_equal(ev[b], ev1)
        np.testing.assert_array_equal(sll[b], sll1)


WORKER = r"""
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["DLAGP_REPO"])
from gpy_dla_detection_amd import parallel
from gpy_dla_detection_amd.synthetic import make_sightline
from tests.test_host_logic import OracleBatch

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
sights = [make_sightline(i, n_unmasked=120 + 4 * i, n_masked=2) for i in range(5)]
out = parallel.model_evidences_sharded(sights, num_samples=24, max_dlas=2, rank=rank, world_size=world,
                                       make_batch=lambda s, device: OracleBatch(s))
np.save(os.path.join(os.environ["DLAGP_OUT"], "rank%d.npy" % rank), out)
dist.barrier()
dist.destroy_process_group()
"""


class OracleBatch:
    """`make_batch` stand-in for the CPU tests: the oracle evaluates, the package drives"""

    def __init__(self, sights):
        from oracle import ref_like as R

        self.R = R
        self.states = [oracle_state(s) for s in sights]

    def eval(self, z, nhi):
        R = self.R
        return np.stack([
            np.array([R.sample_log_likelihood_k_dlas(st, z[b][i], nhi[b][i]) for i in range(z.shape[1])])
            for b, st in enumerate(self.states)
        ])

    def nulls(self):
        return np.array([self.R.null_log_model_evidence(st) for st in self.states])


def test_sharded_evidences_gloo_world_size_2(tmp_path):
    """N > 1 path: contiguous shards (5 sightlines over 2 ranks: 3 + 2), padded all-gather"""
    env = dict(os.environ, DLAGP_REPO=REPO, DLAGP_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1",
               OMP_NUM_THREADS="1")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)]
    res = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    got0 = np.load(tmp_path / "rank0.npy")
    got1 = np.load(tmp_path / "rank1.npy")
    np.testing.assert_array_equal(got0, got1)  # every rank holds the full table
    sights = [make_sightline(i, n_unmasked=120 + 4 * i, n_masked=2) for i in range(5)]
    want = parallel.model_evidences_sharded(sights, num_samples=24, max_dlas=2,
                                            make_batch=lambda s, device: OracleBatch(s))
    assert want.shape == (5, 4) and np.all(np.isfinite(want))
    np.testing.assert_array_equal(got0, want)
