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
from tests.conftest import REPO, free_port, load_golden, oracle_state
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


def test_get_interp_is_shared_between_the_gp_objects_of_a_spectrum():
    """
    The driver's three GP objects per spectrum interpolate the same model onto the same pixels.  The model half is
    produced on demand (first read of this_mu / this_M / this_omega2); the second and third object take the first
    one's result (their own copies, same layout, golden G7 to the last digit), and nothing is shared across different
    pixels, redshifts, learned arrays or in-place edits of the model.
    """
    from gpy_dla_detection_amd import null_gp
    from gpy_dla_detection_amd.null_gp import NullGP

    g = load_golden("set_data.npz")
    z_qso, rest, flux, nv, mask, rest_grid, mu, M, log_omega, lc0, lt0, lb = raw_spectrum_and_model()
    params = Parameters()
    null_gp._INTERP_CACHE.entries.clear()
    gps = [NullGP(params, FixedPrior(), rest_grid, mu, M, log_omega, lc0, lt0, lb) for _ in range(3)]
    assert gps[0]._learned is gps[1]._learned is gps[2]._learned  # one learned model, recognised by content
    assert gps[0].list_M_interpolators is gps[2].list_M_interpolators
    for gp in gps:
        gp.set_data(rest, flux, nv, mask, z_qso, normalize=True, build_model=True)
    assert len(null_gp._INTERP_CACHE.entries) == 0  # nothing interpolated until somebody asks
    for gp in gps:
        for name in ("this_mu", "this_M", "this_omega2"):
            np.testing.assert_allclose(getattr(gp, name), g[name], rtol=1e-14, atol=0, err_msg=name)
            np.testing.assert_array_equal(getattr(gp, name), getattr(gps[0], name))
        assert gp.this_M.flags["F_CONTIGUOUS"]
    assert len(null_gp._INTERP_CACHE.entries) == 1  # one computation, two hits
    assert gps[1].this_M is not gps[0].this_M and gps[1].this_M is gps[1].this_M
    gps[1].this_mu[:] = 0.0  # a caller's edit stays the caller's
    assert gps[0].this_mu[0] != 0.0 and gps[2].this_mu[0] != 0.0
    fresh = NullGP(params, FixedPrior(), rest_grid, mu, M, log_omega, lc0, lt0, lb)
    fresh.set_data(rest, flux, nv, mask, z_qso)
    np.testing.assert_array_equal(fresh.this_mu, gps[0].this_mu)  # the cache's own copy is untouched
    # other pixels / another redshift / another model: computed, not looked up
    other = NullGP(params, FixedPrior(), rest_grid, mu, M, log_omega, lc0, lt0, lb)
    other.set_data(rest, flux, nv, mask, z_qso + 0.01)
    assert not np.array_equal(other.this_mu, fresh.this_mu) and len(null_gp._INTERP_CACHE.entries) == 2
    scaled = NullGP(params, FixedPrior(), rest_grid, 2.0 * mu, M, log_omega, lc0, lt0, lb)
    assert scaled._learned is not fresh._learned
    scaled.set_data(rest, flux, nv, mask, z_qso)
    np.testing.assert_allclose(scaled.this_mu, 2.0 * fresh.this_mu, rtol=1e-15)
    assert len(null_gp._INTERP_CACHE.entries) <= null_gp._INTERP_CACHE.size


def test_in_place_edits_of_the_learned_model_are_noticed_like_the_reference_does():
    """
    The reference's interp1d objects copy the learned arrays when a GP object is constructed (null_gp.py:67-71):
    an edit made in place BEFORE the next construction is a different model for the next object -- also when it
    preserves the sum (a permutation, offsetting changes), which a checksum of sums would miss -- and an edit made
    AFTER an object's construction does not change that object's interpolation.
    """
    from gpy_dla_detection_amd.null_gp import NullGP

    z_qso, rest, flux, nv, mask, rest_grid, mu, M, log_omega, lc0, lt0, lb = raw_spectrum_and_model()
    params = Parameters()
    mu, M, log_omega = mu.copy(), np.array(M, copy=True), log_omega.copy()

    def this_model(*arrays):
        gp = NullGP(params, FixedPrior(), rest_grid, *arrays, lc0, lt0, lb)
        gp.set_data(rest, flux, nv, mask, z_qso)
        return gp, (gp.this_mu.copy(), gp.this_M.copy(), gp.this_omega2.copy())

    first, before = this_model(mu, M, log_omega)
    M[[100, 200]] = M[[200, 100]]  # in place, same array object, same sum
    mu[300] += 0.25
    mu[301] -= 0.25  # offsetting changes
    second, after = this_model(mu, M, log_omega)
    assert second._learned is not first._learned
    assert not np.array_equal(after[1], before[1]) and not np.array_equal(after[0], before[0])
    np.testing.assert_array_equal(after[2], before[2])
    fresh, want = this_model(mu.copy(), M.copy(), log_omega.copy())  # the edited content, brand-new arrays
    assert fresh._learned is second._learned
    for a, b in zip(after, want):
        np.testing.assert_array_equal(a, b)
    # an edit after construction is not seen by the existing object (interp1d copied), a new spectrum included
    M *= 2.0
    second.set_data(rest, flux, nv, mask, z_qso + 0.001)
    ref, want = this_model(mu, 0.5 * M, log_omega)
    ref.set_data(rest, flux, nv, mask, z_qso + 0.001)
    np.testing.assert_array_equal(second.this_M, ref.this_M)
    # rest grid edits count too
    grid2 = rest_grid.copy()
    third, _ = this_model(mu, M, log_omega)
    grid2[5] += 1e-3
    fourth = NullGP(params, FixedPrior(), grid2, mu, M, log_omega, lc0, lt0, lb)
    assert fourth._learned is not third._learned


def test_model_attributes_follow_the_reference_life_cycle():
    from gpy_dla_detection_amd.null_gp import NullGP

    z_qso, rest, flux, nv, mask, rest_grid, mu, M, log_omega, lc0, lt0, lb = raw_spectrum_and_model()
    gp = NullGP(Parameters(), FixedPrior(), rest_grid, mu, M, log_omega, lc0, lt0, lb)
    gp.set_data(rest, flux, nv, mask, z_qso, build_model=False)
    with pytest.raises(AttributeError):  # the reference has not created them either
        gp.this_mu
    gp.get_interp(gp.x, gp.y, gp.this_wavelengths, gp.z_qso)
    want = gp.this_mu.copy()
    gp.set_data(rest, flux, nv, mask, z_qso)
    np.testing.assert_array_equal(gp.this_mu, want)
    gp.this_mu = 2.0 * want  # assignment sticks, the other two are still there
    np.testing.assert_array_equal(gp.this_mu, 2.0 * want)
    assert gp.this_M.shape == (want.shape[0], 20) and not gp._model_on_device_only()
    # pixels outside the learned grid fail in set_data, where interp1d fails in the reference
    narrow = NullGP(Parameters(), FixedPrior(), rest_grid[100:], mu[100:], M[100:], log_omega[100:], lc0, lt0, lb)
    with pytest.raises(ValueError, match="below the interpolation range"):
        narrow.set_data(rest, flux, nv, mask, z_qso)

    class OwnModelHalf(NullGP):
        def get_interp(self, x, y, wavelengths, z_qso):
            self.this_mu, self.this_M, self.this_omega2 = np.ones_like(x), np.zeros((x.shape[0], 20)), np.ones_like(x)

    own = OwnModelHalf(Parameters(), FixedPrior(), rest_grid, mu, M, log_omega, lc0, lt0, lb)
    own.set_data(rest, flux, nv, mask, z_qso)
    assert np.all(own.this_mu == 1.0) and not own._model_on_device_only()


def test_deferred_model_half_uses_the_scalars_of_set_data_like_the_reference():
    """
    The reference bakes log_c_0 / log_tau_0 / log_beta / prev_tau_0 / prev_beta / num_forest_lines and z_qso into
    this_mu, this_M, this_omega2 INSIDE set_data (null_gp.py:157 -> :179-242).  This package defers that half to the
    first evidence call or attribute read, so it must use the values of set_data's moment: changing a scalar afterwards
    changes nothing until the next set_data -- exactly as in the reference.
    """
    from gpy_dla_detection_amd.null_gp import NullGP

    z_qso, rest, flux, nv, mask, rest_grid, mu, M, log_omega, lc0, lt0, lb = raw_spectrum_and_model()
    gp = NullGP(Parameters(), FixedPrior(), rest_grid, mu, M, log_omega, lc0, lt0, lb)
    gp.set_data(rest, flux, nv, mask, z_qso)
    reference_like = NullGP(Parameters(), FixedPrior(), rest_grid, mu, M, log_omega, lc0, lt0, lb)
    reference_like.set_data(rest, flux, nv, mask, z_qso)
    want = [reference_like.this_mu.copy(), reference_like.this_M.copy(), reference_like.this_omega2.copy()]  # built at once
    # between set_data and the first read: every scalar the model half depends on, and the redshift
    gp.log_tau_0 = lt0 + 0.5
    gp.log_c_0 = lc0 - 1.0
    gp.prev_tau_0 = 2.0 * gp.prev_tau_0
    gp.z_qso = z_qso + 0.2
    assert gp._model_pending and gp._model_snapshot[2] == float(z_qso)
    assert gp._model_on_device_only()  # the device route builds from the same snapshot (GPU tests compare the numbers)
    np.testing.assert_array_equal(gp.this_mu, want[0])
    np.testing.assert_array_equal(gp.this_M, want[1])
    np.testing.assert_array_equal(gp.this_omega2, want[2])
    # the next set_data picks the edited scalars up, as the reference's would
    gp.z_qso = z_qso
    gp.set_data(rest, flux, nv, mask, z_qso)
    assert not np.array_equal(gp.this_omega2, want[2]) and not np.array_equal(gp.this_mu, want[0])
    # an explicit get_interp call uses the live attributes, like the reference's
    fresh = NullGP(Parameters(), FixedPrior(), rest_grid, mu, M, log_omega, lc0 - 1.0, lt0 + 0.5, lb,
                   prev_tau_0=gp.prev_tau_0)
    fresh.set_data(rest, flux, nv, mask, z_qso)
    np.testing.assert_array_equal(gp.this_omega2, fresh.this_omega2)


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

# the catalogue driver, 3 spectra over 2 ranks (2 + 1): the per-quasar values of the whole catalogue on every rank after
# ONE all-gather, the per-sample arrays rank-local and written as shards; what every rank allocated and sent is recorded
import json
from gpy_dla_detection_amd import run_bayes_select as rbs
from tests.test_host_logic import FixedPrior, OracleEngine, catalogue_inputs
learned, spectra, z_qsos, params, dla_samples, sub_samples = catalogue_inputs(96)
record = {"gathers": [], "allocated": {}}
plain_gather, plain_empty = parallel.gather_rows, rbs._empty_results
def spying_gather(rows, total, **kw):
    record["gathers"].append([int(np.asarray(rows).shape[0]), int(np.asarray(rows).shape[1]), str(np.asarray(rows).dtype)])
    return plain_gather(rows, total, **kw)
def spying_empty(num, S, max_dlas):
    out = plain_empty(num, S, max_dlas)
    record["allocated"] = {name: list(a.shape) for name, a in out.items()}
    return out
parallel.gather_rows, rbs._empty_results = spying_gather, spying_empty
res = rbs.process_qso(spectra, z_qsos, params, FixedPrior(), dla_samples, sub_samples, learned, max_dlas=3,
                      rank=rank, world_size=world, engine=OracleEngine(learned, params))
record["returned"] = {name: list(np.asarray(a).shape) for name, a in res.items()}
json.dump(record, open(os.path.join(os.environ["DLAGP_OUT"], "catalogue_record%d.json" % rank), "w"))
rbs.write_results(res, os.path.join(os.environ["DLAGP_OUT"], "catalogue.h5"))  # no h5py here: .npz, main file + shards
# ... and with gather_samples every rank holds every array whole (the behaviour up to round 5; int32 rows travel as int32)
record["gathers"] = []
res = rbs.process_qso(spectra, z_qsos, params, FixedPrior(), dla_samples, sub_samples, learned, max_dlas=3,
                      rank=rank, world_size=world, engine=OracleEngine(learned, params), gather_samples=True)
assert ["int32"] == sorted({g[2] for g in record["gathers"] if g[2] != "float64"}), record["gathers"]
parallel.gather_rows, rbs._empty_results = plain_gather, plain_empty
np.savez(os.path.join(os.environ["DLAGP_OUT"], "catalogue_rank%d.npz" % rank), **res)

# ONE sightline, its samples split over the ranks: all-gather of S / world doubles per level, every rank resamples
from gpy_dla_detection_amd.samples import halton
from tests.conftest import load_golden
from tests.test_host_logic import oracle_level_evaluator
g = load_golden("evidences_small.npz")
sight = make_sightline(int(g["sight_index"]), n_unmasked=int(g["sight_n_unmasked"]),
                       n_masked=int(g["sight_n_masked"]), k=int(g["sight_k"]))
S = int(g["S"])
np.random.seed(0)
ev, sll, base = parallel.sweep_sharded_samples(oracle_level_evaluator(sight), g["dla_sample_z"],
                                               10.0 ** (20.0 + 3.0 * halton(S, 3)), 4,
                                               float(g["dla_min_z_separation"]), rank=rank, world_size=world)
np.savez(os.path.join(os.environ["DLAGP_OUT"], "by_sample_rank%d.npz" % rank), ev=ev, sll=sll, base=base)
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


@pytest.mark.parametrize("nproc", [2, 4])
def test_sharded_evidences_gloo_world_size_2(tmp_path, nproc):
    """N > 1 path: cost-balanced shards (5 sightlines over 2 ranks: {0, 1, 4} + {2, 3}), padded all-gather, rows back
    in catalogue order; with 4 ranks the catalogue run (3 spectra) leaves one rank without a spectrum"""
    env = dict(os.environ, DLAGP_REPO=REPO, DLAGP_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1",
               OMP_NUM_THREADS="1")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)]
    res = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    sights = [make_sightline(i, n_unmasked=120 + 4 * i, n_masked=2) for i in range(5)]
    want = parallel.model_evidences_sharded(sights, num_samples=24, max_dlas=2,
                                            make_batch=lambda s, device: OracleBatch(s))
    assert want.shape == (5, 4) and np.all(np.isfinite(want))
    g_cat, g5 = load_golden("catalogue.npz"), load_golden("evidences_small.npz")
    cat0 = np.load(tmp_path / "catalogue_rank0.npz")
    for r in range(nproc):
        np.testing.assert_array_equal(np.load(tmp_path / ("rank%d.npy" % r)), want)  # every rank holds the full table
        # catalogue driver: every rank holds the complete result arrays, equal to the reference's
        cat = np.load(tmp_path / ("catalogue_rank%d.npz" % r))
        for name in cat0.files:
            np.testing.assert_array_equal(cat0[name], cat[name], err_msg=name)
        # by-sample split of one sightline (each rank evaluated its share of the samples of every level): every rank
        # holds the reference's evidences, per-sample likelihoods and -- exactly -- its resampled base indices
        got = np.load(tmp_path / ("by_sample_rank%d.npz" % r))
        np.testing.assert_array_equal(got["base"], g5["dla_base_inds"])
        np.testing.assert_allclose(got["ev"], g5["dla_evidences"], rtol=1e-12)
        np.testing.assert_allclose(got["sll"], g5["dla_sample_ll"], rtol=1e-12, equal_nan=True)
    assert_catalogue_matches_golden(cat0, g_cat, rtol=1e-10)
    # the default: per-sample arrays rank-local.  No rank allocates, returns or sends more than its share of them; the
    # merged shards are the single-process result, array for array
    import json

    from gpy_dla_detection_amd import run_bayes_select as rbs

    per_quasar = 12 + 4 * 3 + 2 * 3 * 3  # doubles per quasar of everything that is not per-sample, max_dlas = 3
    shares = []
    for r in range(nproc):
        record = json.load(open(tmp_path / ("catalogue_record%d.json" % r)))
        share = record["allocated"]["sample_log_likelihoods_dla"][0]
        shares.append(share)
        for name in rbs.PER_SAMPLE:
            assert record["allocated"][name][0] == share and record["returned"][name][0] == share, (r, name)
        assert record["returned"]["local_rows"] == [share] and record["returned"]["log_posteriors_dla"] == [3, 3]
        assert record["gathers"] == [[share, per_quasar, "float64"]]  # ONE collective, 42 doubles per quasar
    assert sum(shares) == 3 and max(shares) <= -(-3 // nproc) + 1
    merged = rbs.merge_results(str(tmp_path / "catalogue.npz"),
                               [str(tmp_path / ("catalogue.rank%d.npz" % r)) for r in range(nproc)])
    assert sorted(merged) == sorted(cat0.files)
    for name in cat0.files:
        np.testing.assert_array_equal(merged[name], cat0[name], err_msg=name)
        assert merged[name].dtype == cat0[name].dtype, name
    with pytest.raises(ValueError, match="cover"):
        rbs.merge_results(str(tmp_path / "catalogue.npz"), [str(tmp_path / "catalogue.rank0.npz")])


# ------------------------------------------------------------------------------------------------
# catalogue driver (f3): run_bayes_select.process_qso against the reference's own loop body (G8)
# ------------------------------------------------------------------------------------------------
class OracleEngine:
    """
    `process_qso` back end for the CPU tests: host `get_interp` of this package + the oracle as
    the evaluator + the host half of every level (`parallel.sweep_batch`).  Checks the driver's
    bookkeeping -- priors, posteriors, MAP, array layout, sharding -- without a GPU.
    """

    def __init__(self, learned, params):
        from gpy_dla_detection_amd.null_gp import NullGP

        self.gp = NullGP(params, None, *learned[:7])
        self.params = params

    def sightline(self, state, z_qso, broadening):
        from oracle import ref_like

        self.gp.get_interp(state["x"], state["y"], state["this_wavelengths"], z_qso)
        return ref_like.SpectrumState(
            y=state["y"], v=state["v"], this_mu=self.gp.this_mu, this_M=self.gp.this_M,
            this_omega2=self.gp.this_omega2, padded_wavelengths=state["padded_wavelengths"],
            unmasked_wavelengths=state["unmasked_wavelengths"],
            keep=~state["pixel_mask"][state["ind_unmasked"]], num_lines=self.params.num_lines, broadening=broadening)

    def batch(self, sightlines):
        batch = OracleBatch([])
        batch.states = list(sightlines)
        return batch

    def nulls(self, batch, sightlines):
        from oracle import ref_like

        return np.array([ref_like.null_log_model_evidence(st) for st in sightlines])

    def sweep(self, batch, sample_z, sample_nhi, max_dlas, min_z_separation, seed):
        return parallel.sweep_batch(batch.eval, sample_z, sample_nhi, max_dlas, min_z_separation, seed)

    def release(self, batch, sightlines):
        pass

    def close(self):
        pass


def catalogue_inputs(S):
    from gpy_dla_detection_amd.run_bayes_select import LearnedModel
    from gpy_dla_detection_amd.samples import DLASamples, SubDLASamples
    from tests.golden_inputs import raw_catalogue

    learned, spectra, z_qsos = raw_catalogue()
    params = Parameters(num_dla_samples=S)
    u = halton(S, 3)
    dla_samples = DLASamples(params, None, halton(S, 2), 20.0 + 3.0 * u)
    sub_samples = SubDLASamples(params, None, halton(S, 2), 19.5 + 0.5 * u, Z_dla=1.0, Z_lls=0.25)
    return LearnedModel(*learned), spectra, z_qsos, params, dla_samples, sub_samples


def assert_catalogue_matches_golden(results, g, rtol):
    S, max_dlas = int(g["S"]), int(g["max_dlas"])
    assert results["sample_log_likelihoods_dla"].shape == (3, S, max_dlas)
    assert results["base_sample_inds"].shape == (3, S, max_dlas - 1) and results["base_sample_inds"].dtype == np.int32
    np.testing.assert_array_equal(results["base_sample_inds"], g["base_sample_inds"])
    for name in ("min_z_dlas", "max_z_dlas", "log_priors_no_dla", "log_priors_lls", "log_priors_dla", "z_qsos"):
        np.testing.assert_allclose(results[name], g[name], rtol=1e-13, err_msg=name)
    for name in ("log_likelihoods_no_dla", "log_likelihoods_lls", "log_likelihoods_dla", "log_posteriors_no_dla",
                 "log_posteriors_lls", "log_posteriors_dla", "sample_log_likelihoods_dla",
                 "sample_log_likelihoods_lls"):
        assert np.array_equal(np.isnan(results[name]), np.isnan(g[name])), name
        np.testing.assert_allclose(results[name], g[name], rtol=rtol, equal_nan=True, err_msg=name)
    np.testing.assert_allclose(results["MAP_z_dlas"], g["MAP_z_dlas"], rtol=1e-13, equal_nan=True)
    np.testing.assert_allclose(results["MAP_log_nhis"], g["MAP_log_nhis"], rtol=1e-13, equal_nan=True)
    np.testing.assert_allclose(results["model_posteriors"], g["model_posteriors"], rtol=1e-6, atol=1e-280)
    np.testing.assert_allclose(results["p_dlas"], g["p_dlas"], rtol=1e-6, atol=1e-280)
    np.testing.assert_allclose(results["p_no_dlas"], g["p_no_dlas"], rtol=1e-6, atol=1e-12)
    assert np.array_equal(np.argmax(results["model_posteriors"], axis=1), np.argmax(g["model_posteriors"], axis=1))
    assert int(results["num_dla_samples"]) == S and int(results["k"]) == 20


class RecordingEngine:
    """records the redshift tables `process_qso` hands to the sweeps; returns empty evidences"""

    def __init__(self):
        self.tables = []

    def sightline(self, state, z_qso, broadening):
        return state["this_wavelengths"]

    def batch(self, sightlines):
        return list(sightlines)

    def model_evidences(self, batch, z_lls, nhi_sub, z_dla, nhi_dla, max_dlas, min_z_separation, seed):
        self.tables.append((z_lls, z_dla))
        nb, S = z_dla.shape
        ll = np.zeros((nb, S, max_dlas))
        return (np.zeros(nb), (np.zeros((nb, 1)), np.zeros((nb, S, 1)), None),
                (np.zeros((nb, max_dlas)), ll, np.zeros((nb, max_dlas - 1, S), dtype=np.int32)))

    def release(self, batch, sightlines):
        pass


def test_batched_sample_redshifts_are_the_holders_own():
    """
    `process_qso` forms the redshift tables of a whole batch in one broadcast expression when the sample holders are the
    stock ones: element for element what `sample_z_dlas` / `sample_z_lls` return per spectrum (reference
    dla_samples.py:94-104, subdla_samples.py:115-125), and ONE table for both models when their offsets coincide.
    """
    from gpy_dla_detection_amd.run_bayes_select import process_qso
    from gpy_dla_detection_amd.samples import DLASamples, SubDLASamples

    S = 257
    learned, spectra, z_qsos, params, dla_samples, sub_samples = catalogue_inputs(S)

    class MyDLASamples(DLASamples):  # any subclass takes the per-spectrum route through the holder's methods
        def sample_z_dlas(self, wavelengths, z_qso):
            return super().sample_z_dlas(wavelengths, z_qso)

    u = halton(S, 3)
    own_route = RecordingEngine()
    process_qso(spectra, z_qsos, params, FixedPrior(), MyDLASamples(params, None, halton(S, 2), 20.0 + 3.0 * u), sub_samples,
                learned, max_dlas=2, batch_size=2, engine=own_route, prefetch=False)
    batched = RecordingEngine()
    process_qso(spectra, z_qsos, params, FixedPrior(), dla_samples, sub_samples, learned, max_dlas=2, batch_size=2,
                engine=batched, prefetch=False)
    other_offsets = SubDLASamples(params, None, halton(S, 5), 19.5 + 0.5 * u, Z_dla=1.0, Z_lls=0.25)
    distinct = RecordingEngine()
    process_qso(spectra, z_qsos, params, FixedPrior(), dla_samples, other_offsets, learned, max_dlas=2, batch_size=2,
                engine=distinct, prefetch=False)
    assert len(batched.tables) == len(own_route.tables) == 2
    for (lls_a, dla_a), (lls_b, dla_b), (lls_c, dla_c) in zip(own_route.tables, batched.tables, distinct.tables):
        assert lls_a is not dla_a
        assert lls_b is dla_b  # same offsets: one table, the library pairs the two models on it
        np.testing.assert_array_equal(dla_b, dla_a)
        np.testing.assert_array_equal(lls_b, lls_a)
        assert lls_c is not dla_c and not np.array_equal(lls_c, dla_c)
        np.testing.assert_array_equal(dla_c, dla_a)
    row = other_offsets.sample_z_lls(params.observed_wavelengths(
        prepare_state(params, spectra[0], z_qsos[0])["x"], z_qsos[0]), z_qsos[0])
    np.testing.assert_array_equal(distinct.tables[0][0][0], row)


def test_prepare_data_shortcuts_are_numpys_arithmetic():
    """
    `prepare_data` forms the normalisation median and the padding of the wavelength grid without NumPy's generic front
    ends (one thread prepares every spectrum of a catalogue): bit for bit np.nanmedian / np.logspace, NaNs, ties, even and
    odd counts, empty windows; `Parameters.z_dla_range` is (min_z_dla, max_z_dla).
    """
    import warnings

    from gpy_dla_detection_amd.null_gp import _logspace, _nanmedian_1d

    rng = np.random.default_rng(11)
    for trial in range(4000):
        n = int(rng.integers(0, 40))
        a = rng.standard_normal(n) * 10.0 ** int(rng.integers(-3, 3))
        if n and trial % 3 == 0:
            a[rng.random(n) < 0.3] = np.nan
        if n and trial % 5 == 0:
            a[rng.integers(0, n)] = a[rng.integers(0, n)]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want, got = np.nanmedian(a), _nanmedian_1d(a)
        assert np.array_equal(np.float64(want), np.float64(got), equal_nan=True), (a, want, got)
    for trial in range(4000):
        lo, spacing, num = rng.uniform(3, 4), 10 ** rng.uniform(-5, -3), int(rng.integers(1, 6))
        for a, b in ((lo - num * spacing, lo - spacing), (lo + spacing, lo + num * spacing), (lo, lo)):
            np.testing.assert_array_equal(_logspace(a, b, num), np.logspace(a, b, num))
    params = Parameters(num_dla_samples=16)
    obs = 10.0 ** (3.56 + 1e-4 * np.arange(4600))
    for z_qso in (2.2, 2.6, 3.3, 4.5):
        for w in (obs, obs[::3], obs[rng.random(obs.size) > 0.05]):
            assert params.z_dla_range(w, z_qso) == (params.min_z_dla(w, z_qso), params.max_z_dla(w, z_qso))


def test_batch_bookkeeping_is_the_per_spectrum_loop():
    """
    `process_qso` turns a batch's evidences into priors / posteriors / MAP values with array expressions over the batch;
    the reference does it spectrum by spectrum (run_bayes_select.py:189-230, bayesian_model_selection.py:68-105,
    dla_gp.py:428-472).  Same values to the last bit, also where a sweep stopped on a NaN evidence.
    """
    from gpy_dla_detection_amd.run_bayes_select import _logsumexp, occupation_log_priors, process_qso

    S, max_dlas = 96, 3
    learned, spectra, z_qsos, params, dla_samples, sub_samples = catalogue_inputs(S)
    spectra, z_qsos = list(spectra) * 4, [z + 0.01 * i for i, z in enumerate(list(z_qsos) * 4)]
    rng = np.random.default_rng(3)

    class ZPrior:
        def less_ind(self, z_qso):
            return 700.0 + 40.0 * z_qso, 9000.0 + z_qso

    class Engine(RecordingEngine):
        def __init__(self):
            super().__init__()
            self.outputs = []

        def model_evidences(self, batch, z_lls, nhi_sub, z_dla, nhi_dla, max_dlas, min_z_separation, seed):
            nb, S = z_dla.shape
            nulls = -500.0 + 30.0 * rng.standard_normal(nb)
            ev_sub = -480.0 + 30.0 * rng.standard_normal((nb, 1))
            ev_dla = -470.0 + 30.0 * rng.standard_normal((nb, max_dlas))
            ll = -470.0 + 30.0 * rng.standard_normal((nb, S, max_dlas))
            ll[rng.random(ll.shape) < 0.05] = np.nan
            if nb > 1:
                ev_dla[1, 2:] = np.nan  # a sweep that stopped after its second level
            base = rng.integers(0, S, size=(nb, max_dlas - 1, S)).astype(np.int32)
            self.tables.append((z_lls, z_dla))
            self.outputs.append((nulls, ev_sub, ev_dla, ll, base))
            return nulls, (ev_sub, ll[:, :, :1].copy(), None), (ev_dla, ll, base)

    engine = Engine()
    res = process_qso(spectra, z_qsos, params, ZPrior(), dla_samples, sub_samples, learned, max_dlas=max_dlas, batch_size=5,
                      engine=engine, prefetch=False)
    i = 0
    for (z_lls, z_dla), (nulls, ev_sub, ev_dla, ll, base) in zip(engine.tables, engine.outputs):
        for b in range(z_dla.shape[0]):
            log_priors = occupation_log_priors(ZPrior(), z_qsos[i], max_dlas, dla_samples, sub_samples)
            log_likelihoods = np.concatenate([[nulls[b]], ev_sub[b], ev_dla[b]])
            log_posteriors = log_likelihoods + log_priors
            np.testing.assert_array_equal(res["log_priors_dla"][i], log_priors[-max_dlas:])
            np.testing.assert_array_equal(res["log_posteriors_dla"][i], log_posteriors[-max_dlas:])
            assert res["log_posteriors_no_dla"][i] == log_posteriors[0] and res["log_likelihoods_lls"][i] == log_likelihoods[1]
            with np.errstate(invalid="ignore"):
                posteriors = np.exp(log_posteriors - _logsumexp(log_posteriors))
            np.testing.assert_array_equal(res["model_posteriors"][i], posteriors)
            p_dla = np.sum(posteriors[-max_dlas:])
            assert np.array_equal(res["p_dlas"][i], p_dla, equal_nan=True)
            assert np.array_equal(res["p_no_dlas"][i], 1 - p_dla, equal_nan=True)
            for level in range(max_dlas):
                top = np.nanargmax(ll[b, :, level])
                members = [top] + [base[b, j, top] for j in range(level)]
                np.testing.assert_array_equal(res["MAP_z_dlas"][i, level, : level + 1], z_dla[b][members])
                np.testing.assert_array_equal(res["MAP_log_nhis"][i, level, : level + 1], dla_samples.log_nhi_samples[members])
                assert np.isnan(res["MAP_z_dlas"][i, level, level + 1:]).all()
            i += 1
    assert i == len(spectra)


def test_pipeline_starts_on_short_batches():
    """
    the pipelined driver begins with a shorter batch (the device waits for the first batch's preparation) -- only as short
    as the launch planner itself keeps single-piece (asked through dlagp_plan_cut: a spectrum's numbers then do not depend on
    the batch it rides in); every spectrum's row lands where the in-order run puts it
    """
    from gpy_dla_detection_amd.run_bayes_select import _single_piece_ramp, process_qso

    assert _single_piece_ramp(32, 10000) == (16,)      # 8 spectra x 625 blocks would be cut in two pieces: 16 it is
    assert _single_piece_ramp(64, 10000) == (16, 32)
    assert _single_piece_ramp(8, 64) == ()             # few samples: every short batch is multi-piece
    S = 10000
    learned, spectra, z_qsos, params, dla_samples, sub_samples = catalogue_inputs(S)
    spectra, z_qsos = list(spectra) * 40, [z + 0.001 * i for i, z in enumerate(list(z_qsos) * 40)]

    class Engine(RecordingEngine):
        def model_evidences(self, batch, z_lls, nhi_sub, z_dla, nhi_dla, max_dlas, min_z_separation, seed):
            out = super().model_evidences(batch, z_lls, nhi_sub, z_dla, nhi_dla, max_dlas, min_z_separation, seed)
            return (z_dla[:, 0].copy(),) + out[1:]  # a value that names the spectrum

    piped, in_order = Engine(), Engine()
    a = process_qso(spectra, z_qsos, params, FixedPrior(), dla_samples, sub_samples, learned, max_dlas=2, batch_size=32,
                    engine=piped)
    b = process_qso(spectra, z_qsos, params, FixedPrior(), dla_samples, sub_samples, learned, max_dlas=2, batch_size=32,
                    engine=in_order, prefetch=False)
    assert sorted(t[1].shape[0] for t in piped.tables) == [8, 16, 32, 32, 32]
    assert [t[1].shape[0] for t in in_order.tables] == [32, 32, 32, 24]
    for name in ("log_likelihoods_no_dla", "min_z_dlas", "max_z_dlas", "sample_log_likelihoods_dla", "base_sample_inds"):
        np.testing.assert_array_equal(a[name], b[name], err_msg=name)
    assert len(set(a["log_likelihoods_no_dla"])) == 120
    # few samples: no short first batches
    learned, few, few_z, params, dla_samples, sub_samples = catalogue_inputs(64)
    few, few_z = list(few) * 10, [z + 0.001 * i for i, z in enumerate(list(few_z) * 10)]
    short = Engine()
    process_qso(few, few_z, params, FixedPrior(), dla_samples, sub_samples, learned, max_dlas=2, batch_size=8, engine=short)
    assert sorted(t[1].shape[0] for t in short.tables) == [6, 8, 8, 8]


def prepare_state(params, spectrum, z_qso):
    from gpy_dla_detection_amd.null_gp import prepare_data

    wavelengths, flux, noise_variance, pixel_mask = spectrum
    return prepare_data(params, params.emitted_wavelengths(wavelengths, z_qso), flux, noise_variance, pixel_mask, z_qso)


def test_process_qso_matches_reference_driver_golden(tmp_path):
    from gpy_dla_detection_amd.run_bayes_select import process_qso, write_results
    from tests.golden_inputs import checksum as crc

    g = load_golden("catalogue.npz")
    learned, spectra, z_qsos, params, dla_samples, sub_samples = catalogue_inputs(int(g["S"]))
    assert crc(*[a for spec in spectra for a in spec[:3]]) == int(g["input_crc"])
    np.random.seed(123)
    before = np.random.get_state()[1].copy()
    results = process_qso(spectra, z_qsos, params, FixedPrior(), dla_samples, sub_samples, learned,
                          max_dlas=int(g["max_dlas"]), batch_size=2, engine=OracleEngine(learned, params),
                          qso_list=["spec-%d" % i for i in range(3)])
    assert np.array_equal(np.random.get_state()[1], before)
    assert_catalogue_matches_golden(results, g, rtol=1e-10)
    in_order = process_qso(spectra, z_qsos, params, FixedPrior(), dla_samples, sub_samples, learned,
                           max_dlas=int(g["max_dlas"]), batch_size=2, engine=OracleEngine(learned, params),
                           prefetch=False)   # no worker thread: same arrays
    for name in results:
        if name != "qso_list":
            np.testing.assert_array_equal(in_order[name], results[name], err_msg=name)
    path = write_results(results, str(tmp_path / "processed_qsos_multi_meanflux.h5"))
    if path.endswith(".npz"):
        back = np.load(path)
        np.testing.assert_array_equal(back["base_sample_inds"], results["base_sample_inds"])
        assert list(back["qso_list"]) == ["spec-0", "spec-1", "spec-2"]


def dr12q_like_kept_pixels(num, rng):
    """
    kept pixels of `num` spectra with a DR12Q-like redshift spread, through the package's own window logic: BOSS
    spectra start at ~3600 A observed, the model covers 911.75-1215.75 A rest (set_parameters.py:33-34), so the
    window is entered at z_qso ~ 2.0 and fully covered from z_qso ~ 2.95 (n = 1250); most quasars sit at z = 2.2-2.6.
    """
    params = Parameters()
    z_qsos = np.clip(2.15 + rng.gamma(2.0, 0.22, size=num), 2.15, 5.5)
    observed = 10.0 ** np.arange(np.log10(3600.0), np.log10(10400.0), 1e-4)  # the BOSS grid: 1e-4 dex pixels
    spectra = []
    for z in z_qsos:
        mask = rng.random(observed.shape[0]) < rng.choice([0.0, 0.02, 0.1])
        spectra.append((observed, None, None, mask))
    from gpy_dla_detection_amd.run_bayes_select import kept_pixel_counts

    return z_qsos, kept_pixel_counts(params, spectra, z_qsos)


def test_cost_balanced_partition_of_a_redshift_ordered_catalogue():
    """
    A spectrum's work follows its kept pixels n, and n follows z_qso through the modelled window (reference
    null_gp.py:139-152, set_parameters.py:125-159).  Equal COUNTS of a redshift-sorted catalogue -- the natural order
    of a catalogue file -- leave the low-z ranks with a fraction of the work; the LPT assignment on the cost model
    balances the ranks to a few parts in a thousand, deterministically.
    """
    rng = np.random.default_rng(7)
    z_qsos, n = dr12q_like_kept_pixels(1024, rng)
    assert n.min() < 400 and n.max() >= 1200  # the window is entered and fully covered within the sample
    order = np.argsort(z_qsos)
    costs = parallel.spectrum_cost(n[order], 10000, 4)
    for world in (2, 4, 8):
        contiguous = parallel.contiguous_assignment(len(costs), world)
        balanced = parallel.balanced_assignment(costs, world)
        assert sorted(np.concatenate(balanced).tolist()) == list(range(len(costs)))  # every spectrum exactly once
        assert all(np.all(np.diff(idx) > 0) for idx in balanced)  # each rank keeps catalogue order
        assert parallel.assignment_imbalance(costs, balanced) <= 1.05
        if world == 8:
            assert parallel.assignment_imbalance(costs, contiguous) > 1.25  # the counter-example
            assert parallel.assignment_imbalance(costs, balanced) <= 1.005
        again = parallel.balanced_assignment(costs.copy(), world)
        assert all(np.array_equal(a, b) for a, b in zip(balanced, again))  # every rank computes the same table
    # few, equal items: ties go to the lowest rank and to catalogue order
    assert [idx.tolist() for idx in parallel.balanced_assignment(np.ones(5), 2)] == [[0, 2, 4], [1, 3]]
    assert [idx.tolist() for idx in parallel.balanced_assignment([], 3)] == [[], [], []]
    with pytest.raises(ValueError):
        parallel.balanced_assignment([1.0, np.nan], 2)


def test_kept_pixel_counts_equal_what_set_data_keeps():
    from gpy_dla_detection_amd.null_gp import prepare_data
    from gpy_dla_detection_amd.run_bayes_select import kept_pixel_counts

    learned, spectra, z_qsos, params, _, _ = catalogue_inputs(8)
    counts = kept_pixel_counts(params, spectra, z_qsos)
    for q, (wavelengths, flux, noise_variance, pixel_mask) in enumerate(spectra):
        rest = params.emitted_wavelengths(wavelengths, z_qsos[q])
        state = prepare_data(params, rest, flux, noise_variance, pixel_mask, z_qsos[q])
        assert counts[q] == state["y"].shape[0]


def test_a_failing_batch_ends_the_catalogue_run_and_frees_every_context():
    """the driver's pipeline: the first failure is re-raised, nothing further starts, every prepared batch is released"""
    from gpy_dla_detection_amd.run_bayes_select import process_qso

    learned, spectra, z_qsos, params, dla_samples, sub_samples = catalogue_inputs(16)
    spectra, z_qsos = spectra * 4, list(z_qsos) * 4  # 12 spectra, batches of 2

    class Flaky(OracleEngine):
        def __init__(self, *a):
            super().__init__(*a)
            self.made, self.released, self.sweeps = 0, 0, 0

        def batch(self, sightlines):
            self.made += 1
            return super().batch(sightlines)

        def sweep(self, *a):
            self.sweeps += 1
            if self.sweeps == 3:  # the DLA sweep of the second batch
                raise RuntimeError("device fell over")
            return super().sweep(*a)

        def release(self, batch, sightlines):
            self.released += 1

    for prefetch in (True, False):
        engine = Flaky(learned, params)
        with pytest.raises(RuntimeError, match="device fell over"):
            process_qso(spectra, z_qsos, params, FixedPrior(), dla_samples, sub_samples, learned, max_dlas=2,
                        batch_size=2, engine=engine, prefetch=prefetch)
        assert engine.made == engine.released and engine.made < 6, (prefetch, engine.made, engine.released)


def test_occupation_log_priors_match_reference_model_selection():
    """the driver's vectorised priors == what BayesModelSelect assembles from the three models (G6)"""
    from gpy_dla_detection_amd.run_bayes_select import occupation_log_priors
    from gpy_dla_detection_amd.samples import DLASamples, SubDLASamples

    g = load_golden("model_selection.npz")
    params = Parameters(num_dla_samples=16)
    u = halton(16, 3)
    dla = DLASamples(params, None, halton(16, 2), 20.0 + 3.0 * u)
    sub = SubDLASamples(params, None, halton(16, 2), 19.5 + 0.5 * u, Z_dla=1.0, Z_lls=0.25)
    got = occupation_log_priors(FixedPrior(), 3.0, 4, dla, sub)
    np.testing.assert_allclose(got, g["log_priors"], rtol=1e-13)
    assert abs(np.exp(got).sum() - 1.0) < 1e-12   # [null, sub-DLA, DLA(1..4)] is a complete partition


def test_prepare_data_is_the_data_half_of_set_data():
    """prepare_data (used by the catalogue driver) leaves exactly what NullGP.set_data leaves (G7)"""
    from gpy_dla_detection_amd.null_gp import prepare_data

    g = load_golden("set_data.npz")
    z_qso, rest, flux, nv, mask, *_ = raw_spectrum_and_model()
    st = prepare_data(Parameters(), rest, flux, nv, mask, z_qso, normalize=True)
    for name in ("x", "y", "v", "padded_wavelengths", "unmasked_wavelengths", "this_wavelengths"):
        np.testing.assert_allclose(st[name], g[name], rtol=1e-14, atol=0, err_msg=name)
    np.testing.assert_array_equal(st["ind_unmasked"], g["ind_unmasked"])
    keep = ~st["pixel_mask"][st["ind_unmasked"]]
    assert keep.sum() == st["y"].shape[0] and keep.shape[0] == st["unmasked_wavelengths"].shape[0]
    raw = prepare_data(Parameters(), rest, flux, nv, mask, z_qso, normalize=False)
    np.testing.assert_allclose(raw["y"] / st["normalization_median"], st["y"], rtol=1e-15)
    assert "normalization_median" not in raw


def test_write_results_hdf5_branch_with_a_stand_in_h5py(tmp_path, monkeypatch):
    """
    `write_results` through its h5py branch (h5py is not installed here): a recording stand-in module checks that
    every dataset of the reference's writer (run_bayes_select.py:248-295) is created, with its shape and dtype.
    """
    import sys
    import types

    from gpy_dla_detection_amd.run_bayes_select import process_qso, write_results

    created = {}

    class File:
        def __init__(self, name, mode):
            assert mode == "w"
            created["__file__"] = name

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

        def create_dataset(self, name, data=None):
            assert name not in created, "dataset written twice: " + name
            created[name] = np.asarray(data)

    fake = types.ModuleType("h5py")
    fake.File = File
    fake.string_dtype = lambda encoding="utf-8": np.dtype("S32")
    monkeypatch.setitem(sys.modules, "h5py", fake)

    g = load_golden("catalogue.npz")
    S, max_dlas = int(g["S"]), int(g["max_dlas"])
    learned, spectra, z_qsos, params, dla_samples, sub_samples = catalogue_inputs(S)
    results = process_qso(spectra, z_qsos, params, FixedPrior(), dla_samples, sub_samples, learned,
                          max_dlas=max_dlas, batch_size=3, engine=OracleEngine(learned, params),
                          qso_list=["spec-%d" % i for i in range(3)])
    target = str(tmp_path / "processed_qsos_multi_meanflux.h5")
    assert write_results(results, target) == target and created.pop("__file__") == target
    N = 3
    reference_datasets = {  # name -> shape, in the reference's order of creation
        "prior_z_qso_increase": (), "k": (), "normalization_min_lambda": (), "normalization_max_lambda": (),
        "min_z_cut": (), "max_z_cut": (), "num_dla_samples": (), "num_lines": (), "num_forest_lines": (),
        "min_z_dlas": (N,), "max_z_dlas": (N,),
        "sample_log_likelihoods_dla": (N, S, max_dlas), "base_sample_inds": (N, S, max_dlas - 1),
        "log_priors_no_dla": (N,), "log_priors_lls": (N,), "log_priors_dla": (N, max_dlas),
        "log_likelihoods_no_dla": (N,), "log_likelihoods_lls": (N,), "log_likelihoods_dla": (N, max_dlas),
        "log_posteriors_no_dla": (N,), "log_posteriors_lls": (N,), "log_posteriors_dla": (N, max_dlas),
        "MAP_z_dlas": (N, max_dlas, max_dlas), "MAP_log_nhis": (N, max_dlas, max_dlas),
        "p_dlas": (N,), "p_no_dlas": (N,), "model_posteriors": (N, 2 + max_dlas),
        "z_qsos": (N,), "qso_list": (N,),
    }
    for name, shape in reference_datasets.items():
        assert name in created, name
        assert created[name].shape == shape, (name, created[name].shape)
    assert created["base_sample_inds"].dtype == np.int32
    assert created["qso_list"].dtype.kind == "S" and created["qso_list"][1] == b"spec-1"
    assert int(created["num_dla_samples"]) == S and int(created["k"]) == params.k
    np.testing.assert_array_equal(created["base_sample_inds"], g["base_sample_inds"])
    # the only dataset beyond the reference's list: the sub-DLA sample likelihoods the driver keeps in memory (:124)
    assert set(created) - set(reference_datasets) == {"sample_log_likelihoods_lls"}


def test_work_decomposition_model():
    """
    dlagp_plan_cut (pure host arithmetic of the MFMA path's planner): how the blocks of 16 samples are cut into
    work items for the 2048 wave slots of the chip.
    """
    import ctypes

    from gpy_dla_detection_amd import _lib

    lib = _lib.load()

    def cut(groups, chunks, forced=0):
        full, frac = ctypes.c_int(), ctypes.c_double()
        _lib.check(lib.dlagp_plan_cut(groups, float(chunks), forced, ctypes.byref(full), ctypes.byref(frac)))
        return full.value, frac.value

    # configs[1]: 625 blocks x 375 chunks -- three full pieces and a remainder that fills the 173 slots left over
    full, frac = cut(625, 375)
    assert full == 3 and 0.30 < frac < 0.32 and 375 * (1 - 3 * frac) >= 7
    # busiest slot: a full piece, against 125 chunks with three equal pieces
    assert frac * 375 < 118
    # large launches: one item per block; forced equal pieces
    assert cut(64 * 625, 375) == (1, 1.0)
    assert cut(625, 375, forced=3) == (3, 1.0 / 3)
    assert cut(625, 375, forced=1000)[0] == 32         # never more than 32 equal pieces, never less than a pass each
    assert cut(625, 60, forced=1000)[0] == 8           # 60 chunks hold 8 passes of 7
    # one rank's slice when ONE sightline's samples are split over 8 GPUs (79 blocks): far more than 8 pieces, so that
    # the 2048 wave slots are filled
    full, frac = cut(79, 375)
    assert full * 79 > 1500 and full * frac <= 1.0 + 1e-12
    assert cut(3, 9) == (1, 1.0)
    # every cut covers the block: full * frac <= 1
    for groups in (1, 7, 300, 625, 1250, 1875, 2047, 2048, 2500, 5000, 10000):
        for chunks in (1, 9, 60, 140, 375, 1000):
            full, frac = cut(groups, chunks)
            assert full >= 1 and 0.0 < frac <= 1.0 and full * frac <= 1.0 + 1e-12, (groups, chunks, full, frac)
    with pytest.raises(ValueError):
        cut(0, 10)
    # A block's piece count travels to the finalisation in a byte: however many wave slots the planner is told to
    # fill and however long the sightline, at most 254 full pieces + a remainder.  The planner options read back.
    value = ctypes.c_int64()
    try:
        _lib.check(lib.dlagp_set_option(b"plan_slots", 1 << 20))
        _lib.check(lib.dlagp_get_option(b"plan_slots", ctypes.byref(value)))
        assert value.value == 1 << 20
        for groups, chunks in ((3, 400000), (1, 1 << 20), (17, 90000), (625, 375)):
            full, frac = cut(groups, chunks)
            assert 1 <= full <= 254 and full * frac <= 1.0 + 1e-12, (groups, chunks, full, frac)
    finally:
        _lib.check(lib.dlagp_set_option(b"plan_slots", 2048))
    for name, default in ((b"plan_slots", 2048), (b"plan_item_cost", 20), (b"plan_rest_first", 0)):
        _lib.check(lib.dlagp_get_option(name, ctypes.byref(value)))
        assert value.value == default, name


def native_prepare_spectrum(lib, params_struct, wavelengths, flux, noise_variance, pixel_mask, z_qso, pad=None):
    """dlagp_prepare_spectrum (host only, no device) -> the dict `prepare_data` returns, plus the four redshift bounds"""
    import ctypes

    from gpy_dla_detection_amd import _lib

    num = wavelengths.shape[0]
    out = {name: np.empty(num + 6) for name in ("x", "y", "v", "this_wavelengths", "unmasked_wavelengths", "padded_wavelengths")}
    keep = np.empty(num, dtype=np.uint8)
    n, n_u, scalars = ctypes.c_int(), ctypes.c_int(), np.empty(7)
    mask8 = np.ascontiguousarray(pixel_mask).view(np.uint8)
    rc = lib.dlagp_prepare_spectrum(
        ctypes.byref(params_struct), num, _lib.ptr(wavelengths), _lib.ptr(flux), _lib.ptr(noise_variance),
        mask8.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), float(z_qso), _lib.ptr(pad) if pad is not None else None,
        ctypes.byref(n), ctypes.byref(n_u), _lib.ptr(scalars), *(_lib.ptr(out[name]) for name in
                                                                  ("x", "y", "v", "this_wavelengths", "unmasked_wavelengths",
                                                                   "padded_wavelengths")),
        keep.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)))
    if rc:
        return rc, None
    sizes = dict(x=n.value, y=n.value, v=n.value, this_wavelengths=n.value, unmasked_wavelengths=n_u.value,
                 padded_wavelengths=n_u.value + 6)
    got = {name: out[name][: sizes[name]] for name in out}
    got.update(keep=keep[: n_u.value].astype(bool), normalization_median=scalars[0], edges=scalars[1:3], bounds=scalars[3:7])
    return 0, got


def same_bits(a, b) -> bool:
    """identical doubles, bit for bit; a NaN matches a NaN whatever its sign (x86's default NaN is the negative one)"""
    a, b = np.ascontiguousarray(a, dtype=np.float64).ravel(), np.ascontiguousarray(b, dtype=np.float64).ravel()
    if a.shape != b.shape:
        return False
    nan = np.isnan(a)
    return bool(np.array_equal(nan, np.isnan(b)) and np.array_equal(a[~nan].view(np.uint64), b[~nan].view(np.uint64)))


def awkward_spectra(rng):
    """spectra that exercise the corners of the data half: NaN flux in the normalisation window, an even and an odd window,
    a window without a valid pixel, masked edges of the modelled window, a short spectrum, unsorted flux ties"""
    obs = 10.0 ** (3.5600 + 1e-4 * np.arange(4600))
    cases = []
    for i, z_qso in enumerate((2.2, 2.5, 2.98, 3.4, 4.9, 2.151)):
        flux = 2.0 + rng.standard_normal(obs.shape[0])
        noise = 0.04 * (1 + rng.random(obs.shape[0]))
        mask = rng.random(obs.shape[0]) < 0.05
        rest = obs / (1 + z_qso)
        window = np.flatnonzero((rest >= 1310.0) & (rest <= 1325.0))
        if i == 0:
            flux[window[::3]] = np.nan
        if i == 1:
            mask[window[1:]] = True            # exactly one valid pixel
        if i == 2:
            mask[window] = True                # none: the median is NaN, and so is everything normalised by it
        if i == 3:
            inside = np.flatnonzero((rest >= 911.75) & (rest <= 1215.75))
            mask[inside[:40]] = True
            mask[inside[-25:]] = True
        if i == 4:
            flux[window] = np.round(flux[window])  # ties
        cases.append(((obs, flux, noise, mask), z_qso))
    short = slice(1800, 2600)
    cases.append(((obs[short].copy(), 2.0 + rng.standard_normal(800), np.full(800, 0.05), np.zeros(800, dtype=bool)), 4.0))
    return cases


def test_native_data_half_is_prepare_data_bit_for_bit():
    """
    dlagp_prepare_spectrum (csrc/prepare_host.h; the per-spectrum body of dlagp_batch_prepare) against
    `null_gp.prepare_data` + `Parameters.z_dla_range` -- the data half of NullGP.set_data, reference null_gp.py:117-177,
    set_parameters.py:125-159 -- on the 2048 spectra of the DR12Q-like mix and on awkward ones: every array and scalar
    identical to the last bit, with the padding pixels from `sightline.numpy_padding` (NumPy's log10 / power, as the
    binding's callback supplies them).  With libm's padding the grid agrees to an ulp -- not always to the bit on hosts
    where NumPy uses SVML, which is why the callback exists.
    """
    import warnings

    from gpy_dla_detection_amd import _lib
    from gpy_dla_detection_amd.null_gp import prepare_data
    from gpy_dla_detection_amd.sightline import numpy_padding, prepare_params
    from gpy_dla_detection_amd.synthetic import make_catalogue

    lib = _lib.load()
    params = Parameters()
    struct = prepare_params(params)
    _, spectra, z_qsos = make_catalogue(2048, mix="dr12q")
    cases = list(zip(spectra, z_qsos)) + awkward_spectra(np.random.default_rng(5))
    libm_differs = 0
    for (wavelengths, flux, noise_variance, pixel_mask), z_qso in cases:
        rest = params.emitted_wavelengths(wavelengths, z_qso)
        with warnings.catch_warnings(), np.errstate(invalid="ignore", divide="ignore"):
            warnings.simplefilter("ignore")  # an empty normalisation window: NumPy warns and answers NaN
            want = prepare_data(params, rest, flux, noise_variance, pixel_mask, z_qso, normalize=True)
        rc, first = native_prepare_spectrum(lib, struct, wavelengths, flux, noise_variance, pixel_mask, z_qso)
        assert rc == 0
        pad = numpy_padding(3, params.pixel_spacing, first["edges"][None, :])[0]
        rc, got = native_prepare_spectrum(lib, struct, wavelengths, flux, noise_variance, pixel_mask, z_qso, pad=pad)
        assert rc == 0
        for name in ("x", "y", "v", "this_wavelengths", "unmasked_wavelengths", "padded_wavelengths"):
            assert got[name].shape == want[name].shape, name
            assert same_bits(got[name], want[name]), name
        assert np.array_equal(got["keep"], ~pixel_mask[want["ind_unmasked"]])
        assert same_bits(got["normalization_median"], want["normalization_median"])
        bounds = params.z_dla_range(want["this_wavelengths"], z_qso) + params.z_dla_range(wavelengths, z_qso)
        assert same_bits(bounds, got["bounds"])
        np.testing.assert_allclose(first["padded_wavelengths"], want["padded_wavelengths"], rtol=3e-16)
        libm_differs += int(first["padded_wavelengths"].tobytes() != got["padded_wavelengths"].tobytes())
    print("spectra whose libm padding differs from NumPy's in some bit: %d of %d" % (libm_differs, len(cases)))
    # what the Python raises comes back as a status: no pixel inside the modelled window
    obs = 10.0 ** (3.5600 + 1e-4 * np.arange(4600))
    rc, _ = native_prepare_spectrum(lib, struct, obs, np.ones(4600), np.ones(4600), np.zeros(4600, dtype=bool), 0.5)
    assert rc == _lib.DLAGP_EINVAL and "modelled window" in _lib.last_error()


def test_native_data_half_matches_reference_golden_set_data():
    """the same entry against golden set G7 -- the reference's own NullGP.set_data (null_gp.py:117-177) on a raw spectrum --
    directly: every array identical (the padding from NumPy's log10 / power, as the binding supplies it)"""
    from gpy_dla_detection_amd import _lib
    from gpy_dla_detection_amd.sightline import numpy_padding, prepare_params

    g = load_golden("set_data.npz")
    z_qso, rest, flux, nv, mask, *_ = raw_spectrum_and_model()
    params = Parameters()
    observed = params.observed_wavelengths(rest, z_qso)  # the native entry starts from read_spec's observed wavelengths
    if not np.array_equal(params.emitted_wavelengths(observed, z_qso), rest):
        pytest.skip("this fixture's rest wavelengths are not the image of an observed grid under / (1 + z)")
    lib, struct = _lib.load(), prepare_params(params)
    rc, first = native_prepare_spectrum(lib, struct, observed, flux, nv, mask, z_qso)
    assert rc == 0
    rc, got = native_prepare_spectrum(lib, struct, observed, flux, nv, mask, z_qso,
                                      pad=numpy_padding(3, params.pixel_spacing, first["edges"][None, :])[0])
    assert rc == 0
    for name in ("x", "y", "v", "padded_wavelengths", "unmasked_wavelengths", "this_wavelengths"):
        assert same_bits(got[name], g[name]), name
    assert same_bits(got["normalization_median"], g["normalization_median"])
    assert same_bits(got["bounds"][:2], [g["min_z_dla"], g["max_z_dla"]])
    assert np.array_equal(got["keep"], ~mask[g["ind_unmasked"]])


def test_native_sample_tables_are_the_holders_own():
    """z_min + (z_max - z_min) * offsets in C (prepare_host.h sample_redshifts) == samples.DLASamples.sample_z_dlas"""
    import ctypes

    from gpy_dla_detection_amd import _lib
    from gpy_dla_detection_amd.samples import DLASamples

    params = Parameters(num_dla_samples=777)
    holder = DLASamples(params, None, halton(777, 2), 20.0 + 3.0 * halton(777, 3))
    plan = ctypes.CDLL(os.path.join(REPO, "gpy_dla_detection_amd", "csrc", "libdlagp_hip.so"))
    assert plan  # (the tables themselves come back from dlagp_batch_prepare: GPU test; here the formula's operation order)
    obs = 10.0 ** (3.5600 + 1e-4 * np.arange(4600))
    for z_qso in (2.2, 2.9, 3.7):
        rest = obs / (1 + z_qso)
        this = (rest * (1 + z_qso))[(rest >= params.min_lambda) & (rest <= params.max_lambda)]
        z_min, z_max = params.z_dla_range(this, z_qso)
        want = holder.sample_z_dlas(this, z_qso)
        width = z_max - z_min
        got = np.array([z_min + width * o for o in holder.offset_samples])
        assert same_bits(got, want)
