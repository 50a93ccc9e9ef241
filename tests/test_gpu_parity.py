"""
Parity of the HIP path (through the C ABI, via ctypes) with the reference's golden vectors
and with the oracle on freshly seeded inputs.  Needs a real MI355X: `pytest -m gpu`.

Tolerances.  BASELINE.json asks for <= 1e-6 relative in log-likelihood and identical argmax.
The device path sums in a different order and uses its own Faddeeva evaluation (2.7e-14 from
wofz), so differences of ~1e-12 are expected; the tests pin RTOL_LL = 1e-9 to catch regressions
long before the contractual 1e-6.
"""
import numpy as np
import pytest

from gpy_dla_detection_amd.samples import halton
from gpy_dla_detection_amd.synthetic import make_qmc_samples, make_sightline
from tests.conftest import load_golden, oracle_state, report_parity
from tests.golden_inputs import MVNPDF_SHAPES, checksum, mvnpdf_inputs
from tests.oracle_fanout import oracle_log_likelihoods

pytestmark = pytest.mark.gpu

RTOL_LL = 1e-9     # log-likelihoods / evidences
RTOL_ABS = 1e-11   # absorption profile values (absolute, profile in [0, 1])


def device_sightline(sight, **kw):
    from gpy_dla_detection_amd.sightline import DeviceSightline

    return DeviceSightline.from_state(sight, **kw)


# ------------------------------------------------------------------------------------------------
# Voigt profile + instrument convolution (reference voigt.py:251-322), via dlagp_absorption
# ------------------------------------------------------------------------------------------------
def test_absorption_matches_reference_golden():
    from gpy_dla_detection_amd.sightline import DeviceSightline

    g = load_golden("voigt_absorption.npz")
    cache = {}
    worst = 0.0
    for i in range(g["case_z"].shape[0]):
        grid = g["grid_" + str(g["case_grid"][i])]
        L, broad = int(g["case_num_lines"][i]), bool(g["case_broadening"][i])
        key = (str(g["case_grid"][i]), L, broad)
        if key not in cache:
            n_u = grid.shape[0] - 6 if broad else grid.shape[0]
            ones = np.ones(n_u)
            cache[key] = DeviceSightline(ones, ones, np.ones((n_u, 2)), ones, ones, grid, np.ones(n_u, bool),
                                         num_lines=L, broadening=broad)
        got = cache[key].absorption([g["case_z"][i]], [10.0 ** g["case_log_nhi"][i]])
        want = g["out_%d" % i]
        worst = max(worst, np.max(np.abs(got - want)))
        np.testing.assert_allclose(got, want, rtol=0, atol=RTOL_ABS)
    assert worst < RTOL_ABS


def test_absorption_of_several_absorbers_is_the_product():
    sight = make_sightline(0, n_unmasked=500, n_masked=13)
    dev = device_sightline(sight)
    z, _, nhi = make_qmc_samples(sight, 6)
    single = [dev.absorption([z[j]], [nhi[j]]) for j in range(3)]
    np.testing.assert_allclose(dev.absorption(z[:3], nhi[:3]), single[0] * single[1] * single[2], rtol=1e-14)
    from oracle import ref_like as R

    want = R.voigt_absorption(sight.padded_wavelengths, nhi[4], z[4])[sight.keep]
    np.testing.assert_allclose(dev.absorption([z[4]], [nhi[4]]), want, rtol=0, atol=RTOL_ABS)


# ------------------------------------------------------------------------------------------------
# low-rank Gaussian log-density (reference null_gp.py:307-360), via dlagp_mvnpdf
# ------------------------------------------------------------------------------------------------
def test_log_mvnpdf_reference_known_answers():
    # reference tests/test_model.py:52-72, integer inputs, tolerance 1e-4 there
    from scipy.stats import multivariate_normal

    from gpy_dla_detection_amd.null_gp import NullGP

    g = load_golden("mvnpdf.npz")
    mu = np.array([1, 2])
    M = np.array([[2, 3, 1], [1, 2, 4]])
    rv = multivariate_normal(mu, np.matmul(M, M.T) + np.eye(2) * 2)
    for y, want in zip(([1, 2], [2, 3], [100, 100]), g["kat_out"]):
        got = NullGP.log_mvnpdf_low_rank(np.array(y), mu, M, np.ones(2) * 2)
        assert abs(got - np.log(rv.pdf(y))) < 1e-4 or rv.pdf(y) == 0
        assert abs(got - want) <= RTOL_LL * abs(want)


@pytest.mark.parametrize("idx", range(len(MVNPDF_SHAPES)))
def test_log_mvnpdf_random_shapes(idx):
    from gpy_dla_detection_amd.sightline import log_mvnpdf_low_rank

    g = load_golden("mvnpdf.npz")
    n, k = MVNPDF_SHAPES[idx]
    y, mu, M, d = mvnpdf_inputs(idx, n, k)
    assert checksum(y, mu, M, d) == int(g["rand_input_crc"][idx])
    want = g["rand_out"][idx, 0]
    for layout in (M, np.asfortranarray(M)):  # the reference's this_M is Fortran-ordered
        got = log_mvnpdf_low_rank(y, mu, layout, d)
        assert abs(got - want) <= RTOL_LL * abs(want)


def test_log_mvnpdf_not_positive_definite_raises_linalgerror():
    from gpy_dla_detection_amd.sightline import log_mvnpdf_low_rank

    y, mu, M, d = mvnpdf_inputs(2, 50, 5)
    with pytest.raises(np.linalg.LinAlgError):
        log_mvnpdf_low_rank(y, mu, M * 10.0, -np.abs(d))  # I + M' D^-1 M indefinite


# ------------------------------------------------------------------------------------------------
# per-sample likelihoods (reference dla_gp.py:311-396)
# ------------------------------------------------------------------------------------------------
def test_sample_log_likelihoods_match_reference_golden():
    g = load_golden("sample_ll.npz")
    sight = make_sightline(int(g["sight_index"]))
    assert checksum(sight.y, sight.this_M, sight.padded_wavelengths, sight.keep) == int(g["sight_crc"])
    dev = device_sightline(sight)
    assert abs(dev.null() - g["null_evidence"]) <= RTOL_LL * abs(g["null_evidence"])
    from gpy_dla_detection_amd.sightline import get_option

    for kd in (1, 2, 3, 4):
        got = dev.eval(g["z_kd%d" % kd], g["nhi_kd%d" % kd])
        # the MFMA kernel with the line cores repaired outside the sample loop: once per wave (1 absorber per
        # sample) or pass by pass (several)
        assert get_option("last_launch_path") == 2 and get_option("last_launch_patch") == (1 if kd == 1 else 2)
        np.testing.assert_allclose(got, g["out_kd%d" % kd], rtol=RTOL_LL)
    got = device_sightline(sight, broadening=False).eval(g["z_kd1"], g["nhi_kd1"])
    np.testing.assert_allclose(got, g["out_nobroad_kd1"], rtol=RTOL_LL)
    got = device_sightline(sight, num_lines=5).eval(g["z_kd2"], g["nhi_kd2"])
    np.testing.assert_allclose(got, g["out_lines5_kd2"], rtol=RTOL_LL)


@pytest.mark.parametrize("shape", [dict(index=5, n_unmasked=401, n_masked=10, k=20),
                                   dict(index=6, n_unmasked=777, n_masked=0, k=7),
                                   dict(index=7, n_unmasked=130, n_masked=40, k=33)])
def test_against_oracle_on_fresh_inputs(shape):
    """seeded inputs that are not in the golden set: ragged n, other ranks, heavy masking"""
    from oracle import ref_like as R

    sight = make_sightline(**shape)
    st = oracle_state(sight)
    dev = device_sightline(sight)
    z, _, nhi = make_qmc_samples(sight, 48)
    rng = np.random.default_rng(3)
    assert abs(dev.null() - R.null_log_model_evidence(st)) <= RTOL_LL * abs(R.null_log_model_evidence(st))
    for kd in (1, 3):
        zz = np.stack([z] + [z[rng.permutation(48)] for _ in range(kd - 1)], axis=1)
        nn = np.stack([nhi] + [nhi[rng.permutation(48)] for _ in range(kd - 1)], axis=1)
        want = np.array([R.sample_log_likelihood_k_dlas(st, zz[i], nn[i]) for i in range(48)])
        np.testing.assert_allclose(dev.eval(zz, nn), want, rtol=RTOL_LL)


@pytest.mark.parametrize("k", [21, 33, 40, 64])
def test_ranks_above_20_run_the_mfma_kernel_in_column_groups(k, restore_options):
    """
    `Parameters(k=...)` is a constructor argument of the reference (set_parameters.py:36).  Ranks 21..64 used to drop to
    the generic kernel (4.7 % of peak); they now run the MFMA kernel once per group of 210 + 20 columns, the groups' sums
    regrouped for the general finalisation: against the oracle, against the generic kernel, one to three absorbers, a
    gapped mask, a batch, and the multi-level device sweep (which recomputes: no profile cache above rank 20).
    """
    from gpy_dla_detection_amd import parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch, get_option
    from oracle import ref_like as R

    set_option = restore_options
    rng = np.random.default_rng(k)
    sights = [make_sightline(40 + k, n_unmasked=900, n_masked=25, k=k), make_sightline(41 + k, n_unmasked=333, n_masked=60, k=k)]
    devs = [device_sightline(s) for s in sights]
    S = 150  # not a multiple of 16
    for sight, dev in zip(sights, devs):
        st = oracle_state(sight)
        z, _, nhi = make_qmc_samples(sight, S)
        for kd in (1, 3):
            zz = np.stack([z] + [z[rng.permutation(S)] for _ in range(kd - 1)], axis=1)
            nn = np.stack([nhi] + [nhi[rng.permutation(S)] for _ in range(kd - 1)], axis=1)
            set_option("path", 2)  # refuse the generic kernel
            got = dev.eval(zz, nn)
            assert get_option("last_launch_path") == 2
            want = np.array([R.sample_log_likelihood_k_dlas(st, zz[i], nn[i]) for i in range(0, S, 7)])
            np.testing.assert_allclose(got[::7], want, rtol=RTOL_LL)
            set_option("path", 1)
            generic = dev.eval(zz, nn)
            assert get_option("last_launch_path") == 1
            np.testing.assert_allclose(got, generic, rtol=1e-10)
            for splits in (1, 3):  # forced cuts: every piece of every column group is added up
                set_option("path", 2)
                set_option("pixel_splits", splits)
                np.testing.assert_allclose(dev.eval(zz, nn), got, rtol=1e-11)
            set_option("pixel_splits", 0)
            set_option("finalize_wave", 0)  # the one-thread-per-sample factorisation: another summation order
            np.testing.assert_allclose(dev.eval(zz, nn), got, rtol=1e-11)
            set_option("finalize_wave", 1)
            set_option("path", 0)
        null = R.null_log_model_evidence(st)
        assert abs(dev.null() - null) <= RTOL_LL * abs(null)
    # a batch of the two, and the occupation sweep on the device against the host levels
    batch = DeviceBatch(devs)
    zs = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    nhi = make_qmc_samples(sights[0], S)[2]
    set_option("path", 2)
    both = batch.eval(zs, np.stack([nhi, nhi]))
    for b, dev in enumerate(devs):
        np.testing.assert_allclose(both[b], dev.eval(zs[b], nhi), rtol=1e-11)  # a batch is cut elsewhere: rounding
    ev, ll, base = parallel.sweep_batch_device(batch, zs, nhi, 3, 0.003, seed=0)
    assert get_option("last_sweep_cached") == 0 and get_option("last_launch_path") == 2
    set_option("path", 0)
    ev_h, ll_h, base_h = parallel.sweep_batch(batch.eval, zs, nhi, 3, 0.003, seed=0)
    np.testing.assert_array_equal(base, base_h)
    np.testing.assert_allclose(ev, ev_h, rtol=1e-11)
    np.testing.assert_allclose(ll, ll_h, rtol=1e-11, equal_nan=True)
    if k == 33:
        # the by-sample split (one rank's slices of every level) and a batch of mixed flavours, both in column groups
        uniforms = parallel.resampling_uniforms(2, 3, S, seed=0)
        sliced = parallel.sweep_sharded_samples_device(batch, zs, nhi, 3, 0.003, uniforms, slices=[(0, 64), (64, 71), (71, S)])
        np.testing.assert_array_equal(sliced[2], base)
        np.testing.assert_allclose(sliced[0], ev, rtol=1e-11)
        flavours = [dict(num_lines=5, broadening=True), dict(num_lines=3, broadening=False)]
        mixed_devs = [device_sightline(s, **f) for s, f in zip(sights, flavours)]
        mixed = DeviceBatch(mixed_devs)
        set_option("path", 2)
        got = mixed.eval(zs, np.stack([nhi, nhi]))
        assert get_option("last_launch_groups") == 2 and get_option("last_launch_path") == 2
        set_option("path", 0)
        for b, dev in enumerate(mixed_devs):
            np.testing.assert_allclose(got[b], dev.eval(zs[b], nhi), rtol=1e-11)
            st = oracle_state(sights[b], **flavours[b])
            want = np.array([R.sample_log_likelihood_k_dlas(st, [zs[b, i]], [nhi[i]]) for i in range(0, S, 11)])
            np.testing.assert_allclose(got[b][::11], want, rtol=RTOL_LL)


# ------------------------------------------------------------------------------------------------
# the occupation sweep through the drop-in classes (reference dla_gp.py:92-225, subdla_gp.py:90-222)
# ------------------------------------------------------------------------------------------------
class FixedPrior:
    def less_ind(self, z_qso):
        return 731.0, 9265.0


def build_models(sight, S):
    """NullGP / SubDLAGP / DLAGP of this package with a synthetic sightline's state installed"""
    from gpy_dla_detection_amd.dla_gp import DLAGP
    from gpy_dla_detection_amd.null_gp import NullGP
    from gpy_dla_detection_amd.samples import DLASamples, SubDLASamples
    from gpy_dla_detection_amd.set_parameters import Parameters
    from gpy_dla_detection_amd.subdla_gp import SubDLAGP

    params = Parameters(num_dla_samples=S, k=sight.k)
    rest = np.linspace(911.75, 1215.75, 5)
    learned = (rest, np.ones(5), np.ones((5, sight.k)), np.zeros(5), 0.0, 0.0, 0.0)
    prior = FixedPrior()
    u = halton(S, 3)
    models = [
        NullGP(params, prior, *learned),
        SubDLAGP(params, prior, SubDLASamples(params, prior, halton(S, 2), 19.5 + 0.5 * u, Z_dla=1.0, Z_lls=0.25), *learned),
        DLAGP(params, prior, DLASamples(params, prior, halton(S, 2), 20.0 + 3.0 * u), *learned),
    ]
    n_u = sight.unmasked_wavelengths.shape[0]
    for gp in models:
        gp.y, gp.v = sight.y, sight.v
        gp.this_mu, gp.this_M, gp.this_omega2 = sight.this_mu, sight.this_M, sight.this_omega2
        gp.padded_wavelengths = sight.padded_wavelengths
        gp.unmasked_wavelengths = sight.unmasked_wavelengths
        gp.this_wavelengths = sight.this_wavelengths
        gp.ind_unmasked = np.ones(n_u, dtype=bool)
        gp.pixel_mask = ~sight.keep
        gp.z_qso = sight.z_qso
    return models


@pytest.mark.parametrize("tag", ["small", "full", "full2k"])
def test_log_model_evidences_match_reference_golden(tag):
    g = load_golden("evidences_%s.npz" % tag)
    kwargs = {k[len("sight_"):]: int(g[k]) for k in g.files if k.startswith("sight_") and k != "sight_crc"}
    sight = make_sightline(**kwargs)
    assert checksum(sight.y, sight.this_M, sight.padded_wavelengths, sight.keep) == int(g["sight_crc"])
    null, sub, dla = build_models(sight, int(g["S"]))

    assert abs(null.log_model_evidence() - g["null_evidence"]) <= RTOL_LL * abs(g["null_evidence"])

    np.random.seed(0)
    ev = dla.log_model_evidences(4)
    np.testing.assert_allclose(dla.dla_samples.sample_z_dlas(dla.this_wavelengths, dla.z_qso), g["dla_sample_z"], rtol=1e-15)
    np.testing.assert_array_equal(dla.base_sample_inds, g["dla_base_inds"])
    assert dla.base_sample_inds.dtype == np.int32
    np.testing.assert_allclose(ev, g["dla_evidences"], rtol=RTOL_LL)
    assert np.array_equal(np.isnan(dla.sample_log_likelihoods), np.isnan(g["dla_sample_ll"]))
    np.testing.assert_allclose(dla.sample_log_likelihoods, g["dla_sample_ll"], rtol=RTOL_LL, equal_nan=True)
    assert int(np.argmax(ev)) == int(np.argmax(g["dla_evidences"]))
    MAP_z, MAP_l = dla.maximum_a_posteriori()
    np.testing.assert_allclose(MAP_z, g["dla_map_z"], rtol=1e-14, equal_nan=True)
    np.testing.assert_allclose(MAP_l, g["dla_map_log_nhi"], rtol=1e-14, equal_nan=True)

    np.random.seed(0)
    ev_sub = sub.log_model_evidences(1)
    np.testing.assert_allclose(ev_sub, g["sub_evidences"], rtol=RTOL_LL)
    np.testing.assert_allclose(sub.sample_log_likelihoods, g["sub_sample_ll"], rtol=RTOL_LL)


def test_model_selection_matches_reference_golden():
    from gpy_dla_detection_amd.bayesian_model_selection import BayesModelSelect

    g = load_golden("model_selection.npz")
    sight = make_sightline(1, n_unmasked=400, n_masked=9)
    assert checksum(sight.y, sight.this_M) == int(g["sight_crc"])
    models = build_models(sight, int(g["S"]))
    bayes = BayesModelSelect([0, 1, 4], 2)
    np.random.seed(0)
    log_post = bayes.model_selection(models, sight.z_qso)
    np.testing.assert_allclose(bayes.log_priors, g["log_priors"], rtol=1e-13)
    np.testing.assert_allclose(bayes.log_likelihoods, g["log_likelihoods"], rtol=RTOL_LL)
    np.testing.assert_allclose(log_post, g["log_posteriors"], rtol=RTOL_LL)
    assert int(np.argmax(log_post)) == int(np.argmax(g["log_posteriors"]))  # identical model selection
    np.testing.assert_allclose(bayes.model_posteriors, g["model_posteriors"], rtol=1e-6, atol=1e-12)
    assert abs(bayes.p_dla - g["p_dla"]) < 1e-9


def test_set_data_pipeline_matches_reference_golden():
    """f1 row: NullGP.set_data / get_interp on the host, then the null evidence on the device"""
    from gpy_dla_detection_amd.null_gp import NullGP
    from gpy_dla_detection_amd.set_parameters import Parameters
    from tests.golden_inputs import raw_spectrum_and_model

    g = load_golden("set_data.npz")
    z_qso, rest, flux, nv, mask, rest_grid, mu, M, log_omega, lc0, lt0, lb = raw_spectrum_and_model()
    gp = NullGP(Parameters(), FixedPrior(), rest_grid, mu, M, log_omega, lc0, lt0, lb)
    gp.set_data(rest, flux, nv, mask, z_qso, normalize=True, build_model=True)
    assert abs(gp.log_model_evidence() - g["null_evidence"]) <= RTOL_LL * abs(g["null_evidence"])


# ------------------------------------------------------------------------------------------------
# interface behaviour and edge cases
# ------------------------------------------------------------------------------------------------
def test_edge_cases_and_errors():
    sight = make_sightline(3, n_unmasked=203, n_masked=4, k=20)  # n = 199, not a multiple of 4
    dev = device_sightline(sight)
    z, _, nhi = make_qmc_samples(sight, 5)
    assert dev.eval(z[:0], nhi[:0]).shape == (0,)                 # empty sweep
    one = dev.eval(z[:1], nhi[:1])
    assert one.shape == (1,) and np.isfinite(one[0])
    # a vanishing column density leaves the null model (to convolution rounding)
    assert abs(dev.eval(z[:1], [0.0])[0] - dev.null()) <= 1e-12 * abs(dev.null())
    # saturated absorber: a = 0 on part of the spectrum, still finite
    assert np.isfinite(dev.eval(z[:1], [1e23])[0])
    with pytest.raises(ValueError):
        dev.eval(z[:3], nhi[:2])                                   # assert at reference dla_gp.py:321
    with pytest.raises(ValueError):
        dev.eval(np.zeros((2, 9)), np.zeros((2, 9)))               # more absorbers than supported
    from gpy_dla_detection_amd.sightline import DeviceSightline

    with pytest.raises(ValueError):                                # mask selects a different number of pixels
        DeviceSightline(sight.y, sight.this_mu, sight.this_M, sight.this_omega2, sight.v,
                        sight.padded_wavelengths, np.ones_like(sight.keep))
    # NaN handling follows the reference: a NaN redshift is dropped by np.nansum (voigt.py:309),
    # leaving the null model; a NaN column density propagates to a NaN likelihood
    assert abs(dev.eval([np.nan], [1e20])[0] - dev.null()) <= 1e-12 * abs(dev.null())
    assert np.isnan(dev.eval(z[:1], [np.nan])[0])


def test_extreme_shapes_against_the_oracle():
    """
    Shapes at the ends of what the boundary accepts, each against the oracle: one kept pixel, fewer pixels than a chunk,
    every pixel but the ends masked, rank 1 and rank 64, eight absorbers per sample with 31 lines (too large for the MFMA
    kernel's per-sample constants: generic kernel), and 100 000 samples of one sightline in a single launch.
    """
    from gpy_dla_detection_amd.sightline import DeviceSightline, get_option
    from oracle import ref_like as R

    rng = np.random.default_rng(5)

    def tiny(n_u, keep, k):
        wl = 3900.0 * 10.0 ** (1e-4 * np.arange(-3, n_u + 3))
        n = int(keep.sum())
        M = 0.1 * rng.standard_normal((n, k))
        return dict(y=1.0 + 0.1 * rng.standard_normal(n), this_mu=np.ones(n), this_M=M, this_omega2=0.01 * np.ones(n),
                    v=0.02 * np.ones(n), padded_wavelengths=wl, unmasked_wavelengths=wl[3:-3], keep=keep)

    cases = [
        tiny(1, np.ones(1, bool), 20),
        tiny(5, np.ones(5, bool), 20),
        tiny(40, np.r_[True, np.zeros(38, bool), True], 20),
        tiny(33, rng.random(33) > 0.3, 1),
        tiny(70, rng.random(70) > 0.1, 64),
    ]
    for case in cases:
        st = R.SpectrumState(**case)
        dev = DeviceSightline(case["y"], case["this_mu"], case["this_M"], case["this_omega2"], case["v"],
                              case["padded_wavelengths"], case["keep"])
        z = np.array([2.2100, 2.2113, 2.2127])  # Lyman alpha inside and next to the few pixels
        nhi = np.array([1e20, 3e21, 1e22])
        got = dev.eval(z, nhi)
        want = np.array([R.sample_log_likelihood_k_dlas(st, [z[i]], [nhi[i]]) for i in range(3)])
        np.testing.assert_allclose(got, want, rtol=RTOL_LL, atol=1e-12)
        got2 = dev.eval(np.stack([z, z[::-1]], 1), np.stack([nhi, nhi[::-1]], 1))
        want2 = np.array([R.sample_log_likelihood_k_dlas(st, [z[i], z[2 - i]], [nhi[i], nhi[2 - i]]) for i in range(3)])
        np.testing.assert_allclose(got2, want2, rtol=RTOL_LL, atol=1e-12)
        null = R.null_log_model_evidence(st)
        assert abs(dev.null() - null) <= RTOL_LL * abs(null) + 1e-12
        dev.close()
    # eight absorbers x 31 lines
    sight = make_sightline(8, n_unmasked=300, n_masked=9)
    dev = device_sightline(sight, num_lines=31)
    st = oracle_state(sight, num_lines=31)
    z, _, nhi = make_qmc_samples(sight, 8)
    z8, n8 = np.stack([np.roll(z, j) for j in range(8)], 1), np.stack([np.roll(nhi, j) for j in range(8)], 1)
    got = dev.eval(z8, n8)
    want = np.array([R.sample_log_likelihood_k_dlas(st, z8[i], n8[i]) for i in range(8)])
    np.testing.assert_allclose(got, want, rtol=RTOL_LL)
    # 100 000 samples in one launch (6250 blocks of 16: beyond the 16-lane finalisation's range)
    sight = make_sightline(9, n_unmasked=420, n_masked=11)
    dev = device_sightline(sight)
    st = oracle_state(sight)
    S = 100000
    z, _, nhi = make_qmc_samples(sight, S)
    got = dev.eval(z, nhi)
    assert get_option("last_launch_path") == 2 and np.all(np.isfinite(got))
    pick = [0, 1, 15, 16, 4097, 65535, 65536, S - 1]
    want = np.array([R.sample_log_likelihood_k_dlas(st, [z[i]], [nhi[i]]) for i in pick])
    np.testing.assert_allclose(got[pick], want, rtol=RTOL_LL)
    # a smaller launch cuts the pixels differently (another summation order): equal to rounding
    np.testing.assert_allclose(got[:640], dev.eval(z[:640], nhi[:640]), rtol=1e-13)


def test_batch_equals_single_sightline_launches():
    from gpy_dla_detection_amd.sightline import DeviceBatch

    sights = [make_sightline(i, n_unmasked=n_u, n_masked=m) for i, (n_u, m) in enumerate([(300, 5), (411, 0), (257, 30)])]
    devs = [device_sightline(s) for s in sights]
    S = 40
    zs, ns = [], []
    for s in sights:
        z, _, nhi = make_qmc_samples(s, S)
        zs.append(np.stack([z, z[::-1]], axis=1))
        ns.append(np.stack([nhi, nhi[::-1]], axis=1))
    batch = DeviceBatch(devs)
    np.testing.assert_allclose(batch.nulls(), [d.null() for d in devs], rtol=1e-14)  # dlagp_batch_null
    got = batch.eval(np.stack(zs), np.stack(ns))
    for b, dev in enumerate(devs):
        # the batch and the single launch may choose different pixel-split counts, i.e. a
        # different (deterministic) summation order: equal to rounding, not bit for bit
        np.testing.assert_allclose(got[b], dev.eval(zs[b], ns[b]), rtol=1e-13)


def test_batch_of_mixed_flavours_runs_as_homogeneous_mfma_groups(restore_options):
    """
    A batch whose members differ in line count, broadening or Lyman-limit flag used to send ALL of them to the
    generic kernel (4.7 % of peak).  It now runs as one MFMA launch group per flavour, every kernel reading samples
    and writing results at the member's position in the caller's arrays: same numbers as member-by-member launches
    (per-sample likelihoods with one and two absorbers, the multi-level device sweep with and without the profile
    cache, the by-sample slices), MFMA kernel throughout, and a non-PD member of ANY group still surfaces as LinAlgError.
    """
    from gpy_dla_detection_amd import parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch, DeviceSightline, get_option, set_option

    sights = [make_sightline(i, n_unmasked=n_u, n_masked=m) for i, (n_u, m) in
              enumerate([(300, 5), (411, 0), (257, 30), (333, 7), (290, 3), (350, 11)])]
    flavours = [dict(), dict(broadening=False), dict(num_lines=5), dict(lls_break=True), dict(), dict(broadening=False)]
    devs = [device_sightline(s, **f) for s, f in zip(sights, flavours)]
    S = 150
    zs, ns = [], []
    for s in sights:
        z, _, nhi = make_qmc_samples(s, S)
        zs.append(z)
        ns.append(nhi)
    zs, ns = np.stack(zs), np.stack(ns)
    batch = DeviceBatch(devs)
    set_option("path", 2)  # refuse the generic kernel
    got1 = batch.eval(zs, ns)
    assert get_option("last_launch_groups") == 4 and get_option("last_launch_path") == 2
    z2, n2 = np.stack([zs, zs[:, ::-1]], axis=2), np.stack([ns, ns[:, ::-1]], axis=2)
    got2 = batch.eval(z2, n2)
    set_option("path", 0)
    np.testing.assert_allclose(batch.nulls(), [d.null() for d in devs], rtol=1e-14)
    for b, dev in enumerate(devs):
        np.testing.assert_allclose(got1[b], dev.eval(zs[b], ns[b]), rtol=1e-13)
        np.testing.assert_allclose(got2[b], dev.eval(z2[b], n2[b]), rtol=1e-13)
    # the generic kernel agrees (independent evaluation of every flavour)
    set_option("path", 1)
    np.testing.assert_allclose(batch.eval(zs, ns), got1, rtol=1e-10)
    set_option("path", 0)
    # multi-level sweep on the device: equal to each member swept alone, cache on and off
    max_dlas, min_sep = 3, 0.01
    uniforms = parallel.resampling_uniforms(len(devs), max_dlas, S, seed=0)
    for cache in (1, 0):
        set_option("profile_cache", cache)
        ev, ll, base = batch.sweep(zs, ns, max_dlas, min_sep, uniforms)
        assert get_option("last_launch_groups") == 4
        # the two 3-line flavours and the Lyman-limit one keep level-1 profiles; 5 lines x 3 absorbers recomputes
        for b, dev in enumerate(devs):
            alone = DeviceBatch([dev])
            ev1, ll1, base1 = alone.sweep(zs[b:b + 1], ns[b:b + 1], max_dlas, min_sep, uniforms[b:b + 1])
            alone.close()
            np.testing.assert_array_equal(base[b], base1[0])
            np.testing.assert_allclose(ev[b], ev1[0], rtol=1e-12)
            np.testing.assert_allclose(ll[b], ll1[0], rtol=1e-12, equal_nan=True)
    set_option("profile_cache", 1)
    # by-sample slices of the mixed batch (one rank's share of every level)
    got = parallel.sweep_sharded_samples_device(batch, zs, ns, max_dlas, min_sep, uniforms,
                                                slices=[(0, 64), (64, 71), (71, S)])
    np.testing.assert_array_equal(got[2], base)
    np.testing.assert_allclose(got[0], ev, rtol=1e-12)
    batch.close()
    # a member whose B = I + M' D^-1 M is indefinite, in a launch group that is not the last: the later ones must not clear the flag
    bad = sights[4]
    set_option("negative_noise_mfma", 1)  # keep the invalid member on the MFMA kernel: its non-PD flag is under test
    neg = DeviceSightline(bad.y, bad.this_mu, 10.0 * bad.this_M, bad.this_omega2, -np.abs(bad.v), bad.padded_wavelengths,
                          bad.keep)
    set_option("negative_noise_mfma", 0)
    mixed = DeviceBatch([neg, devs[1], devs[2]])
    with pytest.raises(np.linalg.LinAlgError):
        mixed.eval(zs[[4, 1, 2]], ns[[4, 1, 2]])
    assert get_option("last_launch_groups") == 3
    with pytest.raises(np.linalg.LinAlgError):
        mixed.sweep(zs[[4, 1, 2]], ns[[4, 1, 2]], 2, min_sep, uniforms[:3, :1])
    mixed.close()


def test_model_evidences_in_one_call_equal_the_three_calls(restore_options):
    """
    dlagp_batch_model_evidences = dlagp_batch_null + dlagp_batch_sweep (sub-DLA model, one level) + dlagp_batch_sweep
    (DLA model) enqueued back to back and waited for once: the same arrays bit for bit -- homogeneous batch and batch of
    mixed flavours, profile cache on and off, from two host threads at once (on their own streams, and with option
    shared_work_stream: kernels of both threads on one stream, each call waiting for its own completion event only) --
    and a non-PD member in any of the three models is reported.
    """
    import threading
    import time

    from gpy_dla_detection_amd import _lib, parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch, DeviceSightline, get_option, set_option

    sights = [make_sightline(20 + i, n_unmasked=n_u, n_masked=m) for i, (n_u, m) in
              enumerate([(300, 5), (411, 0), (257, 30), (333, 7)])]
    S, max_dlas, min_sep = 210, 3, 0.004
    zs = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    z_sub = np.stack([make_qmc_samples(s, S, sub_dla=True)[0] for s in sights])
    nhi = make_qmc_samples(sights[0], S)[2]
    nhi_sub = make_qmc_samples(sights[0], S, sub_dla=True)[2]
    uniforms = parallel.resampling_uniforms(len(sights), max_dlas, S, seed=3)

    def check(batch, rows=slice(None)):
        nulls = batch.nulls()
        sub = batch.sweep(z_sub[rows], nhi_sub, 1, min_sep)
        dla = batch.sweep(zs[rows], nhi, max_dlas, min_sep, uniforms[rows])
        # the three launches kept apart (option pair_models = 0, set for this whole test): bit for bit
        got_nulls, got_sub, got_dla = batch.model_evidences(z_sub[rows], nhi_sub, zs[rows], nhi, max_dlas, min_sep,
                                                            uniforms[rows])
        np.testing.assert_array_equal(got_nulls, nulls)
        for got, want in ((got_sub, sub), (got_dla, dla)):
            assert got[2].shape == want[2].shape and got[1].shape == want[1].shape
            for a, b in zip(got, want):
                np.testing.assert_array_equal(a, b)  # NaN == NaN here
        # the default: both models' one-absorber level in ONE launch when their redshift samples are equal (they are:
        # the synthetic sub-DLA samples share the DLA samples' offsets, as the reference's loaders do).  Every sample's
        # sums are formed as before, but the launch of 2 S rows is cut at other pixel positions: agreement to rounding.
        set_option("pair_models", 1)
        pair_nulls, pair_sub, pair_dla = batch.model_evidences(z_sub[rows], nhi_sub, zs[rows], nhi, max_dlas, min_sep,
                                                               uniforms[rows])
        set_option("pair_models", 0)
        np.testing.assert_array_equal(pair_nulls, nulls)
        for got, want in ((pair_sub, sub), (pair_dla, dla)):
            np.testing.assert_allclose(got[0], want[0], rtol=1e-12, equal_nan=True)
            np.testing.assert_allclose(got[1], want[1], rtol=1e-12, equal_nan=True)
            np.testing.assert_array_equal(got[2], want[2])  # the resampled base indices
        return got_nulls, got_sub, got_dla

    assert get_option("shared_work_stream") == 1  # the default since round 5: one kernel stream per device
    set_option("pair_models", 0)  # this test is about the call structure; the pair launch is switched on where it is compared
    flavours = [dict(), dict(broadening=False), dict(num_lines=5), dict(lls_break=True)]
    for kinds in ([dict()] * 4, flavours):
        devs = [device_sightline(s, **f) for s, f in zip(sights, kinds)]
        batch = DeviceBatch(devs)
        for cache in (1, 0):
            set_option("profile_cache", cache)
            first = check(batch)
        set_option("profile_cache", 1)
        # per-sightline column densities: (nb, S) tables for both models
        wide = batch.model_evidences(z_sub, np.tile(nhi_sub, (4, 1)), zs, np.tile(nhi, (4, 1)), max_dlas, min_sep, uniforms)
        np.testing.assert_array_equal(wide[2][1], first[2][1])
        np.testing.assert_array_equal(wide[1][0], first[1][0])
        batch.close()
        for d in devs:
            d.close()

    # the smallest shapes: one level (no uniforms, no base indices), one sample
    devs = [device_sightline(s) for s in sights[:2]]
    small = DeviceBatch(devs)
    for S1, levels in ((S, 1), (1, 2), (17, 1)):
        u1 = parallel.resampling_uniforms(2, levels, S1, seed=5)
        got_nulls, got_sub, got_dla = small.model_evidences(z_sub[:2, :S1], nhi_sub[:S1], zs[:2, :S1], nhi[:S1], levels, min_sep,
                                                            u1 if levels > 1 else None)
        np.testing.assert_array_equal(got_nulls, small.nulls())
        want_dla = small.sweep(zs[:2, :S1], nhi[:S1], levels, min_sep, u1 if levels > 1 else None)
        for a, b in zip(got_dla, want_dla):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(got_sub[1], small.sweep(z_sub[:2, :S1], nhi_sub[:S1], 1, min_sep)[1])
        assert got_dla[2].shape == (2, levels - 1, S1)
    small.close()
    for d in devs:
        d.close()

    # two host threads, a batch each, many calls: every call's results are its own
    devs = [device_sightline(s) for s in sights]
    halves = [DeviceBatch(devs[:2]), DeviceBatch(devs[2:])]
    want = [check(halves[0], slice(0, 2)), check(halves[1], slice(2, 4))]
    errors = []

    def worker(i):
        rows = slice(2 * i, 2 * i + 2)
        try:
            for _ in range(6):
                got = halves[i].model_evidences(z_sub[rows], nhi_sub, zs[rows], nhi, max_dlas, min_sep, uniforms[rows])
                np.testing.assert_array_equal(got[0], want[i][0])
                np.testing.assert_array_equal(got[1][1], want[i][1][1])
                np.testing.assert_array_equal(got[2][1], want[i][2][1])
                np.testing.assert_array_equal(got[2][2], want[i][2][2])
                np.testing.assert_array_equal(halves[i].nulls(), want[i][0])
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    # the call in two phases (the drop-in classes do host work in between): the same arrays, also with a first phase that
    # is never finished in front of it
    halves[0].model_evidences_start(z_sub[:2], nhi_sub, zs[:2], nhi, max_dlas, min_sep)
    halves[0].model_evidences_start(z_sub[:2], nhi_sub, zs[:2], nhi, max_dlas, min_sep)
    time.sleep(0.01)  # "host work"
    two = halves[0].model_evidences_finish(uniforms[:2])
    np.testing.assert_array_equal(two[0], want[0][0])
    for a, b in zip(two[1], want[0][1]):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(two[2], want[0][2]):
        np.testing.assert_array_equal(a, b)
    with pytest.raises(ValueError):
        _lib.check(_lib.load().dlagp_batch_model_evidences_finish(halves[0]._handle, None, None, None, None, None, None, None))

    for shared in (0, 1):
        set_option("shared_work_stream", shared)
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
    set_option("shared_work_stream", 1)
    # every host-pointer call returns with its own work complete: the kernel stream has nothing queued now
    # (the read-back tools/measure_catalogue.py polls to see how often a catalogue run leaves the device without work)
    assert get_option("work_stream_idle") == 1
    for h in halves:
        h.close()

    # a member with an indefinite B: whichever model meets it first, the call reports it
    bad = sights[0]
    set_option("negative_noise_mfma", 1)
    neg = DeviceSightline(bad.y, bad.this_mu, 10.0 * bad.this_M, bad.this_omega2, -np.abs(bad.v), bad.padded_wavelengths,
                          bad.keep)
    set_option("negative_noise_mfma", 0)
    mixed = DeviceBatch([devs[1], neg])
    with pytest.raises(np.linalg.LinAlgError):
        mixed.model_evidences(z_sub[[1, 0]], nhi_sub, zs[[1, 0]], nhi, max_dlas, min_sep, uniforms[:2])
    mixed.close()
    with pytest.raises(ValueError):
        DeviceBatch(devs[:2]).model_evidences(z_sub[:2, :50], nhi_sub[:50], zs[:2], nhi, max_dlas, min_sep, uniforms[:2])


def test_cached_levels_read_a_shared_row_once(restore_options):
    """
    The resampled base indices concentrate on very few QMC points, so the 16 samples of a wave mostly stack the SAME
    cached row; such a wave reads it once per pass (option shared_rows, default 1) instead of 16 times.  Same arithmetic:
    the sweep is bit-identical with the option off, on sightlines whose levels hold both kinds of waves.
    """
    from gpy_dla_detection_amd import parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch, get_option, set_option

    sights = [make_sightline(30 + i, n_unmasked=420 + 40 * i, n_masked=6) for i in range(3)]
    devs = [device_sightline(s) for s in sights]
    batch = DeviceBatch(devs)
    S, max_dlas = 1000, 4  # S not a multiple of 16: the last wave repeats its last sample
    zs = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    nhi = make_qmc_samples(sights[0], S)[2]
    uniforms = parallel.resampling_uniforms(3, max_dlas, S, seed=2)
    assert get_option("shared_rows") == 1
    got = batch.sweep(zs, nhi, max_dlas, 0.002, uniforms)
    assert get_option("last_sweep_cached") == 1
    set_option("shared_rows", 0)
    want = batch.sweep(zs, nhi, max_dlas, 0.002, uniforms)
    set_option("shared_rows", 1)
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)
    blocks = got[2][:, :, : S // 16 * 16].reshape(3, max_dlas - 1, -1, 16)
    shared = (blocks == blocks[..., :1]).all(axis=-1)
    assert shared.any() and not shared.all()  # both kinds of waves ran
    batch.close()


def test_cached_levels_match_recomputed_levels_for_any_cut(restore_options):
    """
    The profile cache is tiled (16 points x 64 slots) and a cached level walks a work item's chunk range tile by tile --
    the first and last pass of an item cover part of a tile when the launch plan cuts the sightline elsewhere.  Same
    chunks, same steps, same products in the same order: whatever the cut (1 ... 9 pieces per block, rows that end
    inside a tile, S not a multiple of 16, sightlines of different lengths in one batch) the cached sweep equals the
    sweep that recomputes every level from the Voigt profiles, bit for bit.
    """
    from gpy_dla_detection_amd import parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch, get_option, set_option

    sights = [make_sightline(50, n_unmasked=1001, n_masked=9), make_sightline(51, n_unmasked=77, n_masked=2),
              make_sightline(52, n_unmasked=640, n_masked=0)]
    devs = [device_sightline(s) for s in sights]
    batch = DeviceBatch(devs)
    S, max_dlas = 203, 4
    zs = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    nhi = make_qmc_samples(sights[0], S)[2]
    uniforms = parallel.resampling_uniforms(3, max_dlas, S, seed=5)
    for splits in (0, 1, 2, 3, 5, 9):
        set_option("pixel_splits", splits)
        set_option("profile_cache", 1)
        got = batch.sweep(zs, nhi, max_dlas, 0.002, uniforms)
        assert get_option("last_sweep_cached") == 1
        set_option("profile_cache", 0)
        want = batch.sweep(zs, nhi, max_dlas, 0.002, uniforms)
        assert get_option("last_sweep_cached") == 0
        for a, b in zip(got, want):
            np.testing.assert_array_equal(a, b, err_msg="pixel_splits=%d" % splits)
    batch.close()


def test_sweep_workspaces_are_recycled_between_batches():
    """
    A destroyed batch leaves its sweep workspace (partial sums, profile cache: buffers above the pool's size limit) to the
    next batch created: a sequence of batches of the same shape allocates its big buffers once and never frees one (a
    free synchronises the device) -- read back through the `big_allocs` / `big_frees` diagnostics.
    """
    from gpy_dla_detection_amd import parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch, get_option

    sights = [make_sightline(40 + i, n_unmasked=900, n_masked=4) for i in range(4)]
    S = 6000
    zs = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    nhi = make_qmc_samples(sights[0], S)[2]
    uniforms = parallel.resampling_uniforms(4, 3, S, seed=1)

    def one_batch():
        devs = [device_sightline(s) for s in sights]
        batch = DeviceBatch(devs)
        out = batch.sweep(zs, nhi, 3, 0.003, uniforms)
        batch.close()
        for d in devs:
            d.close()
        return out

    first = one_batch()  # whatever this one had to allocate
    allocs, frees = get_option("big_allocs"), get_option("big_frees")
    for _ in range(3):
        again = one_batch()
        np.testing.assert_array_equal(again[0], first[0])
    assert get_option("big_allocs") == allocs and get_option("big_frees") == frees


def test_full_size_properties():
    """BASELINE config 2 (n = 3000, k = 20, S = 10 000): size-independent properties"""
    sight = make_sightline(1)
    dev = device_sightline(sight)
    S = 10000
    z, log_nhi, nhi = make_qmc_samples(sight, S)
    ll = dev.eval(z, nhi)
    assert ll.shape == (S,) and np.all(np.isfinite(ll))
    # (1) determinism: the same sweep twice is identical bit for bit (pieces are added in a fixed order);
    # order independence: a permuted sweep returns the permuted results to rounding (a sample's pixel range is
    # cut where its block of 16 samples is cut, so the summation order depends on the block it lands in)
    np.testing.assert_array_equal(dev.eval(z, nhi), ll)
    perm = np.random.default_rng(0).permutation(S)
    np.testing.assert_allclose(dev.eval(z[perm], nhi[perm]), ll[perm], rtol=1e-13)
    # (2) the best sample sits on the injected absorber
    best = int(np.argmax(ll))
    assert abs(z[best] - sight.injected_z[0]) < 0.02 and abs(log_nhi[best] - sight.injected_log_nhi[0]) < 0.4
    # (3) absorbers commute; a second absorber of zero column density changes nothing
    sub = slice(0, 512)
    two = dev.eval(np.stack([z[sub], z[sub][::-1]], 1), np.stack([nhi[sub], nhi[sub][::-1]], 1))
    swapped = dev.eval(np.stack([z[sub][::-1], z[sub]], 1), np.stack([nhi[sub][::-1], nhi[sub]], 1))
    np.testing.assert_allclose(two, swapped, rtol=1e-12)
    ghost = dev.eval(np.stack([z[sub], z[sub][::-1]], 1), np.stack([nhi[sub], np.zeros(512)], 1))
    np.testing.assert_allclose(ghost, ll[sub], rtol=1e-12)
    # (4) the WHOLE population against the oracle: every one of the 10 000 samples (reference dla_gp.py:132-159 ->
    # :311-329), fanned over the host's cores; the same best sample
    want = oracle_log_likelihoods([(sight, z, nhi)])[0]
    worst = float(np.max(np.abs(ll - want) / np.abs(want)))
    report_parity("configs[1]: 1 sightline x 10 000 samples, n = 3000", worst, S)
    assert worst <= RTOL_LL, worst
    assert int(np.argmax(ll)) == int(np.argmax(want))


# ------------------------------------------------------------------------------------------------
# the two evaluation paths and the pixel-split decomposition
# ------------------------------------------------------------------------------------------------
@pytest.fixture
def restore_options():
    from gpy_dla_detection_amd.sightline import set_option

    yield set_option
    set_option("path", 0)
    set_option("pixel_splits", 0)
    set_option("profile_cache", 1)
    set_option("core_patch", 1)
    set_option("shared_work_stream", 1)
    set_option("shared_rows", 1)
    set_option("negative_noise_mfma", 0)
    set_option("finalize_wave", 1)
    set_option("fuse_finalize", 1)
    set_option("pair_models", 1)
    set_option("level_spread", 1)
    set_option("level_search_groups", 16)
    set_option("fold_evidence", 1)


@pytest.mark.parametrize("shape", [dict(index=4, n_unmasked=1250, n_masked=28, k=20),
                                   dict(index=8, n_unmasked=333, n_masked=77, k=12),
                                   dict(index=9, n_unmasked=70, n_masked=3, k=20)])
def test_generic_and_mfma_paths_agree(shape, restore_options):
    """path 1 = one workgroup per sample (VALU), path 2 = MFMA kernel; any split count"""
    set_option = restore_options
    sight = make_sightline(**shape)
    dev = device_sightline(sight)
    S = 203  # not a multiple of the 64-sample workgroup
    z, _, nhi = make_qmc_samples(sight, S)
    zz = np.stack([z, z[::-1], np.roll(z, 7)], axis=1)
    nn = np.stack([nhi, nhi[::-1], np.roll(nhi, 7)], axis=1)
    for kd in (1, 3):
        set_option("path", 1)
        want = dev.eval(zz[:, :kd], nn[:, :kd])
        set_option("path", 2)
        for splits in (1, 2, 3, 8, 1000):
            set_option("pixel_splits", splits)
            got = dev.eval(zz[:, :kd], nn[:, :kd])
            np.testing.assert_allclose(got, want, rtol=1e-11, err_msg="kd=%d splits=%d" % (kd, splits))
        set_option("pixel_splits", 0)


def test_mfma_path_other_line_counts(restore_options):
    """1, 5 and all 31 series members on the MFMA kernel: against the oracle, and against the generic kernel"""
    from oracle import ref_like as R

    set_option = restore_options
    sight = make_sightline(2, n_unmasked=600, n_masked=11)
    z, _, nhi = make_qmc_samples(sight, 96)
    for lines in (1, 5, 31):
        dev = device_sightline(sight, num_lines=lines)
        st = oracle_state(sight, num_lines=lines)
        pick = np.arange(0, 96, 4)
        want = np.array([R.sample_log_likelihood_k_dlas(st, [z[i]], [nhi[i]]) for i in pick])
        set_option("path", 2)
        got = dev.eval(z, nhi)
        np.testing.assert_allclose(got[pick], want, rtol=RTOL_LL, err_msg="%d lines" % lines)
        set_option("path", 1)
        np.testing.assert_allclose(got, dev.eval(z, nhi), rtol=1e-11)
    # two absorbers x 31 lines on the MFMA kernel
    dev = device_sightline(sight, num_lines=31)
    st = oracle_state(sight, num_lines=31)
    z2, n2 = np.stack([z, z[::-1]], axis=1), np.stack([nhi, nhi[::-1]], axis=1)
    set_option("path", 2)
    got = dev.eval(z2, n2)
    want = np.array([R.sample_log_likelihood_k_dlas(st, z2[i], n2[i]) for i in range(0, 96, 12)])
    np.testing.assert_allclose(got[::12], want, rtol=RTOL_LL)


def test_device_voigt_profile_matches_reference_golden():
    """
    The device Faddeeva evaluation itself (not end-to-end): both forms the kernels use, for all 31 series members,
    against the reference's voigt.Voigt values on golden grid G1 (801 velocities from the line centre to |x| ~ 4e4
    Doppler widths).  The measured maxima go to profiles/r02_faddeeva_accuracy.txt.
    """
    import os

    from gpy_dla_detection_amd import lyman_series as ls
    from gpy_dla_detection_amd.sightline import debug_voigt

    g = load_golden("voigt_profile.npz")
    v, want = g["velocity"], g["profile"]
    x = np.abs(v) / (np.sqrt(2) * ls.SIGMA_CGS)
    core, wing = x < 16.0, x >= 16.0
    assert core.sum() > 100 and wing.sum() > 100
    worst = {}
    for form, name in ((0, "generic"), (1, "mfma")):
        for line in range(31):
            got = debug_voigt(form, line, v)
            rel = np.abs(got - want[line]) / np.abs(want[line])
            for region, sel in (("core", core), ("wing", wing)):
                key = (name, region)
                worst[key] = max(worst.get(key, 0.0), float(rel[sel].max()))
    lines = ["max relative difference of the device Voigt profile to the reference's (scipy.special.wofz), golden grid G1,",
             "31 lines x 801 velocities; core = |x| < 16 Doppler widths (Taylor table, both forms), wing = |x| >= 16"]
    for key in sorted(worst):
        lines.append("  %-8s %-5s %.2e" % (key[0], key[1], worst[key]))
    print("\n".join(lines))  # pytest -s shows it; DLAGP_REPORT_DIR=<dir> keeps it (profiles/r02_faddeeva_accuracy.txt)
    if os.environ.get("DLAGP_REPORT_DIR"):
        with open(os.path.join(os.environ["DLAGP_REPORT_DIR"], "faddeeva_accuracy.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    assert worst[("generic", "core")] < 2e-13 and worst[("mfma", "core")] < 2e-13, worst
    assert worst[("generic", "wing")] < 1e-13, worst
    assert worst[("mfma", "wing")] < 2e-12, worst   # degree-4 fit of the wing, DESIGN.md section 6


def test_level_evidence_kernel_matches_numpy():
    """device log-mean-exp with the reference's NaN semantics (dla_gp.py:155-190)"""
    import warnings

    from gpy_dla_detection_amd import sightline as sl
    from tests.conftest import HipBuffers

    sl.set_option("timing", 0)  # loads the library, and with it the HIP runtime HipBuffers binds to
    rng = np.random.default_rng(5)
    nb, S = 3, 5000
    ll = rng.normal(-800.0, 30.0, size=(nb, S))
    mask = rng.random((nb, S)) < 0.1
    mask[2, :] = True  # an all-masked level must come out NaN
    bufs = HipBuffers()
    try:
        for level in (0, 2):
            want_ll = ll - np.log(S)
            want_ll[mask] = np.nan
            with warnings.catch_warnings(), np.errstate(all="ignore"):
                warnings.simplefilter("ignore")
                top = np.nanmax(want_ll, axis=1)
                want = top + np.log(np.nanmean(np.exp(want_ll - top[:, None]), axis=1)) - np.log(S) * level
            d_ll = bufs.to_device(ll)
            d_mask = bufs.to_device(mask.astype(np.uint8))
            d_ev = bufs.to_device(np.zeros(nb))
            sl.level_evidence_device(d_ll, nb, S, level, d_ev, d_too_close=d_mask)
            got = bufs.to_host(d_ev, (nb,), np.float64)
            np.testing.assert_allclose(got[:2], want[:2], rtol=1e-13)
            assert np.isnan(got[2]) and np.isnan(want[2])
            np.testing.assert_allclose(bufs.to_host(d_ll, (nb, S), np.float64), want_ll, rtol=1e-15, equal_nan=True)
    finally:
        bufs.free()


def test_sharded_driver_on_device_matches_oracle_driver():
    """parallel.model_evidences_sharded through DeviceBatch (single process) vs the oracle evaluator"""
    from gpy_dla_detection_amd import parallel
    from tests.test_host_logic import OracleBatch

    sights = [make_sightline(i, n_unmasked=150 + 10 * i, n_masked=4) for i in range(3)]
    got = parallel.model_evidences_sharded(sights, num_samples=64, max_dlas=3)
    want = parallel.model_evidences_sharded(sights, num_samples=64, max_dlas=3,
                                            make_batch=lambda s, device: OracleBatch(s))
    assert got.shape == (3, 5)
    np.testing.assert_allclose(got, want, rtol=RTOL_LL)
    assert np.array_equal(np.argmax(got, axis=1), np.argmax(want, axis=1))  # identical model selection


from tests.golden_inputs import GAPPED_CASES, gapped_sightline  # noqa: E402


def test_ragged_mask_runs_on_the_mfma_path(restore_options):
    """masked runs longer than one 64-entry pass are planned with zero-weight slots: MFMA kernel, oracle parity"""
    from oracle import ref_like as R

    set_option = restore_options
    for gaps in (((203, 333),), ((10, 75), (76, 140), (300, 520)), ((0, 64), (500, 597))):
        sight = gapped_sightline(gaps=gaps)
        dev = device_sightline(sight)
        z, _, nhi = make_qmc_samples(sight, 40)
        st = oracle_state(sight)
        want = np.array([R.sample_log_likelihood_k_dlas(st, [z[i]], [nhi[i]]) for i in range(40)])
        set_option("path", 2)  # must not raise: every mask is plannable
        for splits in (0, 1, 3):
            set_option("pixel_splits", splits)
            np.testing.assert_allclose(dev.eval(z, nhi), want, rtol=RTOL_LL)
        set_option("pixel_splits", 0)
        z2, n2 = np.stack([z, z[::-1]], axis=1), np.stack([nhi, nhi[::-1]], axis=1)
        want2 = np.array([R.sample_log_likelihood_k_dlas(st, z2[i], n2[i]) for i in range(0, 40, 5)])
        np.testing.assert_allclose(dev.eval(z2, n2)[::5], want2, rtol=RTOL_LL)
        set_option("path", 0)
        assert abs(dev.null() - R.null_log_model_evidence(st)) <= RTOL_LL * abs(R.null_log_model_evidence(st))


def test_gapped_masks_match_reference_golden(restore_options):
    """golden set G11: the reference itself on sightlines with long masked runs -- per-sample likelihoods with one and
    two absorbers, the null evidence and a seeded three-level sweep -- against the MFMA kernel (path = 2)"""
    from gpy_dla_detection_amd import parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch

    set_option = restore_options
    g = load_golden("gapped.npz")
    for c, case in enumerate(GAPPED_CASES):
        sight = gapped_sightline(**case)
        assert checksum(sight.y, sight.this_M, sight.padded_wavelengths, sight.keep) == int(g["crc_%d" % c])
        dev = device_sightline(sight)
        set_option("path", 2)
        np.testing.assert_allclose(dev.eval(g["z1_%d" % c], g["n1_%d" % c]), g["out1_%d" % c], rtol=RTOL_LL)
        np.testing.assert_allclose(dev.eval(g["z2_%d" % c], g["n2_%d" % c]), g["out2_%d" % c], rtol=RTOL_LL)
        set_option("path", 0)
        assert abs(dev.null() - g["null_%d" % c]) <= RTOL_LL * abs(g["null_%d" % c])
        S = g["sweep_z_%d" % c].shape[0]
        batch = DeviceBatch([dev])
        set_option("path", 2)
        ev, ll, base = batch.sweep(g["sweep_z_%d" % c][None, :], g["sweep_nhi_%d" % c], 3, float(g["min_z_separation"]),
                                   parallel.resampling_uniforms(1, 3, S, seed=0))
        set_option("path", 0)
        np.testing.assert_array_equal(base[0], g["sweep_base_%d" % c])
        np.testing.assert_allclose(ev[0], g["sweep_ev_%d" % c], rtol=RTOL_LL)
        np.testing.assert_allclose(ll[0], g["sweep_ll_%d" % c], rtol=RTOL_LL, equal_nan=True)


def test_one_gapped_sightline_does_not_slow_the_batch_down(restore_options):
    """one gapped + 15 clean sightlines in one DeviceBatch.sweep: MFMA kernel and profile cache stay on"""
    from gpy_dla_detection_amd import parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch, get_option

    set_option = restore_options
    sights = [make_sightline(i, n_unmasked=420 + 8 * i, n_masked=5) for i in range(15)]
    sights.insert(7, gapped_sightline(index=21, n_unmasked=500, gaps=((100, 230), (301, 369))))
    devs = [device_sightline(s) for s in sights]
    batch = DeviceBatch(devs)
    S, max_dlas = 320, 3
    z = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    nhi = make_qmc_samples(sights[0], S)[2]
    set_option("path", 2)  # the generic kernel is not allowed to serve any launch of this sweep
    ev, ll, base = parallel.sweep_batch_device(batch, z, nhi, max_dlas, 0.01)
    assert get_option("last_sweep_cached") == 1
    set_option("path", 0)
    want_ev, want_ll, want_base = parallel.sweep_batch(batch.eval, z, nhi, max_dlas, 0.01)
    np.testing.assert_allclose(ev, want_ev, rtol=1e-11)
    np.testing.assert_allclose(ll, want_ll, rtol=1e-11, equal_nan=True)
    assert np.array_equal(base, want_base)
    # and the gapped member against the oracle
    from oracle import ref_like as R

    st = oracle_state(sights[7])
    pick = np.arange(3, S, 37)
    want = np.array([R.sample_log_likelihood_k_dlas(st, [z[7, i]], [nhi[i]]) for i in pick]) - np.log(S)
    np.testing.assert_allclose(ll[7, pick, 0], want, rtol=RTOL_LL)


# ------------------------------------------------------------------------------------------------
# BASELINE configs[2..4] at their one-GPU widths (S = 10 000, n = 3000, k = 20)
# ------------------------------------------------------------------------------------------------
_FULL = {}


def full_size_sightlines(count):
    """the synthetic sightlines 0..count-1 of the bench workload, created once per test session"""
    for i in range(len(_FULL), count):
        _FULL[i] = make_sightline(i)
    return [_FULL[i] for i in range(count)]


def spread(b: int, count: int, S: int) -> np.ndarray:
    """`count` distinct sample indices spread over [0, S), different for every sightline b"""
    return (b * 37 + np.arange(count) * (S // count)) % S


def test_config3_batch_of_64_sightlines_full_size():
    """BASELINE configs[2]: 64 sightlines x 10 000 samples, DLA(1) and sub-DLA, one batch launch each"""
    from gpy_dla_detection_amd.sightline import DeviceBatch

    sights = full_size_sightlines(64)
    devs = [device_sightline(s) for s in sights]
    batch = DeviceBatch(devs)
    S = 10000
    z = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    per = 128  # oracle-checked samples per sightline and model, plus the device's best sample of the sightline
    for sub in (False, True):
        nhi = np.broadcast_to(make_qmc_samples(sights[0], S, sub_dla=sub)[2], z.shape)
        ll = batch.eval(z, nhi)
        assert ll.shape == (64, S) and np.all(np.isfinite(ll))
        # the batch launch (one piece per block) agrees with a single-sightline launch (pieces + remainder)
        np.testing.assert_allclose(ll[5], devs[5].eval(z[5], nhi[5]), rtol=1e-13)
        # EVERY sightline against the oracle (reference dla_gp.py:311-329 / subdla_gp.py:224-242)
        picks = [np.unique(np.append(spread(b, per, S), int(np.argmax(ll[b])))) for b in range(64)]
        wants = oracle_log_likelihoods([(sights[b], z[b, picks[b]], nhi[b, picks[b]]) for b in range(64)])
        worst = 0.0
        for b in range(64):
            got = ll[b, picks[b]]
            worst = max(worst, float(np.max(np.abs(got - wants[b]) / np.abs(wants[b]))))
            assert int(np.argmax(got)) == int(np.argmax(wants[b])), b  # the same best sample among those compared
        report_parity("configs[2]: 64 sightlines, %s model" % ("sub-DLA" if sub else "DLA(1)"), worst,
                      sum(len(p) for p in picks))
        assert worst <= RTOL_LL, worst
    # sightlines with an injected absorber prefer the DLA model at the injected redshift
    nhi = np.broadcast_to(make_qmc_samples(sights[0], S)[2], z.shape)
    ll = batch.eval(z, nhi)
    for b in range(1, 64, 3):  # index % 3 == 1: one injected absorber
        best = int(np.argmax(ll[b]))
        assert abs(z[b, best] - sights[b].injected_z[0]) < 0.02
    batch.close()
    for d in devs:
        d.close()


def test_config4_one_gpu_share_128_sightlines_through_shard_runner():
    """BASELINE configs[3] (1024 spectra / 8 GPUs): one rank's 128 sightlines x [null, sub-DLA, DLA(1)] at S = 10 000"""
    from gpy_dla_detection_amd import parallel
    from oracle import ref_like as R

    sights = full_size_sightlines(128)
    S = 10000
    runner = parallel.ShardRunner(sights, S, max_dlas=1)
    table = runner.run()
    assert table.shape == (128, 3) and np.all(np.isfinite(table))
    z = runner.z
    log_S = np.log(S)
    # the null evidence of every sightline (reference null_gp.py:294-305)
    nulls = np.array([R.null_log_model_evidence(oracle_state(s)) for s in sights])
    worst_null = float(np.max(np.abs(table[:, 0] - nulls) / np.abs(nulls)))
    report_parity("configs[3]: null evidence of 128 sightlines", worst_null, 128)
    assert worst_null <= RTOL_LL
    per = 32
    for col, nhi, name in ((1, runner.nhi_sub, "sub-DLA"), (2, runner.nhi_dla, "DLA(1)")):
        # the evidence is a log-mean-exp over all S samples (dla_gp.py:180-190): recomputed from the device's
        # likelihoods, which the oracle checks on every sightline
        dev_ll = runner.batch.batch.eval(z, np.broadcast_to(nhi, z.shape))
        top = dev_ll.max(axis=1)
        want = top + np.log(np.mean(np.exp(dev_ll - top[:, None]), axis=1)) - log_S  # Occam term of dla_gp.py:157-159
        assert np.all(np.abs(table[:, col] - want) <= 1e-11 * np.abs(want))
        picks = [np.unique(np.append(spread(b, per, S), int(np.argmax(dev_ll[b])))) for b in range(128)]
        wants = oracle_log_likelihoods([(sights[b], z[b, picks[b]], nhi[picks[b]]) for b in range(128)])
        worst = 0.0
        for b in range(128):
            got = dev_ll[b, picks[b]]
            worst = max(worst, float(np.max(np.abs(got - wants[b]) / np.abs(wants[b]))))
            assert int(np.argmax(got)) == int(np.argmax(wants[b])), b
        report_parity("configs[3]: 128 sightlines, %s model" % name, worst, sum(len(p) for p in picks))
        assert worst <= RTOL_LL, worst
    # argmax over [null, DLA(1)] = injected truth (synthetic.py: index % 3 absorbers injected)
    for b in range(128):
        assert (table[b, 2] > table[b, 0]) == (b % 3 != 0), b


def test_config5_one_gpu_share_32_sightlines_four_levels():
    """BASELINE configs[4] (256 spectra / 8 GPUs): one rank's 32 sightlines x DLA(1..4) at S = 10 000, levels on the device"""
    from gpy_dla_detection_amd import parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch, get_option

    sights = full_size_sightlines(32)
    devs = [device_sightline(s) for s in sights]
    batch = DeviceBatch(devs)
    S, max_dlas = 10000, 4
    min_sep = 3000.0e3 / 299792458.0
    z = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    nhi = make_qmc_samples(sights[0], S)[2]
    ev, ll, base = parallel.sweep_batch_device(batch, z, nhi, max_dlas, min_sep)
    assert get_option("last_sweep_cached") == 1 and get_option("last_launch_path") == 2
    assert ev.shape == (32, 4) and ll.shape == (32, S, 4) and base.shape == (32, 3, S)
    assert np.all(np.isfinite(ev))
    # (1) the oracle on every level, 1000 samples per level and sightline (plus the level's best sample): the parameter
    # sets of level l stack the QMC point on the absorbers the DEVICE drew for it (base indices, dla_gp.py:135-150)
    members = (0, 7, 13, 22, 31)
    jobs, where = [], []
    for b in members:
        for level in range(max_dlas):
            col = np.where(np.isnan(ll[b, :, level]), -np.inf, ll[b, :, level])
            pick = np.unique(np.append(spread(b + level, 1000, S), int(np.argmax(col))))
            idx = np.concatenate([pick[None, :], base[b, :level][:, pick]], axis=0)  # (level + 1, len(pick))
            jobs.append((sights[b], z[b][idx].T, nhi[idx].T))
            where.append((b, level, pick, idx))
    wants = oracle_log_likelihoods(jobs)
    worst = np.zeros(max_dlas)
    for (b, level, pick, idx), want in zip(where, wants):
        want = want - np.log(S)
        got = ll[b, pick, level]
        ok = ~np.isnan(got)  # NaN = two absorbers closer than min_z_separation (dla_gp.py:164-177)
        zs = np.sort(z[b][idx], axis=0)
        assert np.array_equal(~ok, np.any(np.diff(zs, axis=0) < min_sep, axis=0))
        worst[level] = max(worst[level], float(np.max(np.abs(got[ok] - want[ok]) / np.abs(want[ok]))))
        assert int(np.argmax(np.where(ok, got, -np.inf))) == int(np.argmax(np.where(ok, want, -np.inf))), (b, level)
    for level in range(max_dlas):
        report_parity("configs[4]: %d sightlines, level %d (%d absorbers per sample)" % (len(members), level + 1, level + 1),
                      worst[level], sum(len(w[2]) for w in where if w[1] == level))
    assert worst.max() <= RTOL_LL, worst
    # (2) device levels == host levels (NumPy half of every level, reference RNG) on two sightlines
    two = DeviceBatch([devs[4], devs[22]])
    want_ev, want_ll, want_base = parallel.sweep_batch(two.eval, z[[4, 22]], nhi, max_dlas, min_sep, seed=0)
    assert np.array_equal(base[[4, 22]], want_base)
    np.testing.assert_allclose(ev[[4, 22]], want_ev, rtol=1e-11)
    np.testing.assert_allclose(ll[[4, 22]], want_ll, rtol=1e-11, equal_nan=True)
    two.close()
    # (3) model selection over [null, DLA(1..4)] picks the number of injected absorbers
    nulls = batch.nulls()
    for b in range(32):
        assert int(np.argmax(np.concatenate([[nulls[b]], ev[b]]))) == b % 3, (b, nulls[b], ev[b])
    batch.close()
    for d in devs:
        d.close()


# ------------------------------------------------------------------------------------------------
# device-resident occupation sweep (dlagp_batch_sweep): all of reference dla_gp.py:92-225 on the GPU
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["small", "full", "full2k"])
def test_device_sweep_matches_reference_golden(tag):
    """same golden vectors as the class-API test, but masking / evidence / resampling run on the device"""
    from gpy_dla_detection_amd import parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch

    g = load_golden("evidences_%s.npz" % tag)
    kwargs = {k[len("sight_"):]: int(g[k]) for k in g.files if k.startswith("sight_") and k != "sight_crc"}
    sight = make_sightline(**kwargs)
    _, sub, dla = build_models(sight, int(g["S"]))
    batch = DeviceBatch([dla.device_sightline()])
    sample_z = dla.dla_samples.sample_z_dlas(dla.this_wavelengths, dla.z_qso)[None, :]

    ev, ll, base = parallel.sweep_batch_device(batch, sample_z, dla.dla_samples.nhi_samples, 4,
                                               dla.min_z_separation, seed=0)
    np.testing.assert_array_equal(base[0], g["dla_base_inds"])  # the reference's np.random.choice draws
    assert base.dtype == np.int32
    np.testing.assert_allclose(ev[0], g["dla_evidences"], rtol=RTOL_LL)
    assert np.array_equal(np.isnan(ll[0]), np.isnan(g["dla_sample_ll"]))
    np.testing.assert_allclose(ll[0], g["dla_sample_ll"], rtol=RTOL_LL, equal_nan=True)
    assert int(np.argmax(ev[0])) == int(np.argmax(g["dla_evidences"]))

    ev_sub, ll_sub, base_sub = parallel.sweep_batch_device(batch, sample_z, sub.dla_samples.nhi_samples, 1,
                                                           sub.min_z_separation, seed=0)
    assert base_sub.shape == (1, 0, int(g["S"]))
    np.testing.assert_allclose(ev_sub[0], g["sub_evidences"], rtol=RTOL_LL)
    np.testing.assert_allclose(ll_sub[0], g["sub_sample_ll"], rtol=RTOL_LL)


@pytest.mark.parametrize("path,cache", [(0, 1), (0, 0), (1, 1)])
def test_device_sweep_equals_host_level_logic(path, cache, restore_options):
    """
    several ragged sightlines; MFMA kernel with and without the profile cache, and the generic kernel
    (which never caches): device levels == host levels
    """
    from gpy_dla_detection_amd import parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch

    restore_options("path", path)
    restore_options("profile_cache", cache)
    sights = [make_sightline(i, n_unmasked=n_u, n_masked=m) for i, (n_u, m) in enumerate([(300, 5), (411, 0), (257, 30)])]
    devs = [device_sightline(s) for s in sights]
    batch = DeviceBatch(devs)
    S = 1200 if path == 0 else 150   # > 1024 exercises the blocked scan with several samples per thread
    zs = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    nhi = make_qmc_samples(sights[0], S)[2]
    min_sep = 3000.0 * 1000.0 / 299792458.0
    np.random.seed(77)
    state = np.random.get_state()[1].copy()
    want = parallel.sweep_batch(batch.eval, zs, nhi, 4, min_sep, seed=3)
    got = parallel.sweep_batch_device(batch, zs, nhi, 4, min_sep, seed=3)
    assert np.array_equal(np.random.get_state()[1], state)  # the caller's RNG stream is left as found
    np.testing.assert_array_equal(got[2], want[2])
    assert got[2].max() > 0 and np.isnan(want[1]).any()  # resampling and the separation mask both happened
    assert np.array_equal(np.isnan(got[1]), np.isnan(want[1]))
    np.testing.assert_allclose(got[1], want[1], rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(got[0], want[0], rtol=1e-12)


def test_device_sweep_with_more_samples_than_fit_in_lds(restore_options):
    """
    level_finish keeps a level's values, weights and cdf in LDS while 8 S bytes fit (S <= 18 000); beyond that the cdf
    lives in global memory (`level_finish<false>`, the only form up to round 3).  S = 20 000 on a short sightline:
    device levels == host levels (indices identical), and the profile cache (tiled, 20 000 rows) == recomputation.
    """
    from gpy_dla_detection_amd import parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch, get_option

    sights = [make_sightline(60, n_unmasked=130, n_masked=3), make_sightline(61, n_unmasked=90, n_masked=0)]
    batch = DeviceBatch([device_sightline(s) for s in sights])
    S, max_dlas = 20000, 3
    zs = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    nhi = make_qmc_samples(sights[0], S)[2]
    min_sep = 3000.0 * 1000.0 / 299792458.0
    want = parallel.sweep_batch(batch.eval, zs, nhi, max_dlas, min_sep, seed=4)
    got = parallel.sweep_batch_device(batch, zs, nhi, max_dlas, min_sep, seed=4)
    assert get_option("last_sweep_cached") == 1
    np.testing.assert_array_equal(got[2], want[2])
    assert got[2].max() > 0
    assert np.array_equal(np.isnan(got[1]), np.isnan(want[1]))
    np.testing.assert_allclose(got[1], want[1], rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(got[0], want[0], rtol=1e-12)
    restore_options("profile_cache", 0)
    again = parallel.sweep_batch_device(batch, zs, nhi, max_dlas, min_sep, seed=4)
    assert get_option("last_sweep_cached") == 0
    for a, b in zip(got, again):
        np.testing.assert_array_equal(a, b)
    batch.close()


@pytest.mark.parametrize("tag", ["small", "full2k"])
def test_by_sample_split_on_device_matches_reference_golden(tag):
    """
    SURVEY 8e by-sample partition through the C ABI (dlagp_batch_level_eval_device + dlagp_level_finish_device):
    every level evaluated in slices of the samples, as G ranks would, then finished on the full vector -- the
    reference's evidences, likelihoods and base indices (golden G5), and bit-identical to the unsliced device sweep.
    """
    import torch

    from gpy_dla_detection_amd import parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch

    torch.cuda.init()
    g = load_golden("evidences_%s.npz" % tag)
    kwargs = {k[len("sight_"):]: int(g[k]) for k in g.files if k.startswith("sight_") and k != "sight_crc"}
    sight = make_sightline(**kwargs)
    S, max_dlas = int(g["S"]), g["dla_evidences"].shape[0]
    nhi = 10.0 ** (20.0 + 3.0 * halton(S, 3))
    z = g["dla_sample_z"][None, :]
    uniforms = parallel.resampling_uniforms(1, max_dlas, S, seed=0)
    batch = DeviceBatch([device_sightline(sight)])
    min_sep = float(g["dla_min_z_separation"])
    results = []
    for slices in (None, [(0, S // 3), (S // 3, S // 3 + 7), (S // 3 + 7, S)]):
        ev, ll, base = parallel.sweep_sharded_samples_device(batch, z, nhi, max_dlas, min_sep, uniforms, slices=slices)
        np.testing.assert_array_equal(base[0], g["dla_base_inds"])
        np.testing.assert_allclose(ev[0], g["dla_evidences"], rtol=RTOL_LL)
        np.testing.assert_allclose(ll[0], g["dla_sample_ll"], rtol=RTOL_LL, equal_nan=True)
        results.append((ev, ll, base))
    want_ev, want_ll, want_base = batch.sweep(z, nhi, max_dlas, min_sep, uniforms)
    assert np.array_equal(results[0][2], want_base) and np.array_equal(results[1][2], want_base)
    np.testing.assert_allclose(results[0][0], want_ev, rtol=1e-12)
    np.testing.assert_allclose(results[1][1], want_ll, rtol=1e-12, equal_nan=True)


@pytest.mark.parametrize("S,nb,max_dlas", [(10000, 1, 4), (1200, 3, 4), (96, 2, 3), (1, 1, 2), (18000, 1, 2), (20000, 2, 3)])
def test_level_finish_spread_over_many_workgroups_equals_one_workgroup_bit_for_bit(S, nb, max_dlas, restore_options):
    """
    The kernels that end a level (Occam term, separation mask, evidence, resampling: reference dla_gp.py:155-218) in
    their two forms -- one 1024-thread workgroup per sightline, or three launches that spread a sightline over many
    compute units, with any number of workgroups sharing the searches: evidences, likelihoods, base indices and near-tie
    counts identical to the last bit.  (S = 20 000 does not fit the LDS: both settings run the one-workgroup kernel.)
    """
    from gpy_dla_detection_amd import parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch, get_option

    sights = [make_sightline(70 + i, n_unmasked=150 + 40 * i, n_masked=3 * i) for i in range(nb)]
    batch = DeviceBatch([device_sightline(s) for s in sights])
    zs = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    nhi = make_qmc_samples(sights[0], S)[2]
    min_sep = 3000.0 * 1000.0 / 299792458.0
    uniforms = parallel.resampling_uniforms(nb, max_dlas, S, seed=5)
    restore_options("level_spread", 0)
    want = batch.sweep(zs, nhi, max_dlas, min_sep, uniforms)
    want_ties = batch.last_ties.copy()
    assert get_option("last_level_spread") == 0
    if S > 1:
        assert want[2].max() > 0  # resampling happened
    for groups in (1, 5, 16, 64):
        restore_options("level_spread", 1)
        restore_options("level_search_groups", groups)
        got = batch.sweep(zs, nhi, max_dlas, min_sep, uniforms)
        assert get_option("last_level_spread") == (1 if S <= 18000 else 0)
        for a, b in zip(got, want):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(batch.last_ties, want_ties)
    batch.close()


def numpy_cdf(level_ll):
    """the cdf legacy np.random.choice searches (reference dla_gp.py:209-218), with NumPy's own roundings"""
    weights = np.exp(level_ll - np.nanmax(level_ll))
    weights[np.isnan(weights)] = 0.0
    cdf = (weights / weights.sum()).cumsum()
    cdf /= cdf[-1]
    return cdf


def planted_uniforms(cdf, rng):
    """uniforms exactly ON cdf steps, one ulp below and one ulp above them, and ordinary ones in between"""
    S = cdf.shape[0]
    steps = cdf[rng.integers(0, S, size=S)]
    kind = rng.integers(0, 4, size=S)
    u = np.where(kind == 0, steps, np.where(kind == 1, np.nextafter(steps, 0.0), np.where(kind == 2, np.nextafter(steps, 2.0),
                                                                                        rng.random(S))))
    return np.clip(u, 0.0, np.nextafter(1.0, 0.0))  # random_sample draws from [0, 1)


@pytest.mark.parametrize("spread", [1, 0])
def test_resampling_draws_numpys_indices_where_uniforms_sit_on_cdf_steps(spread, restore_options):
    """
    `base_sample_inds` are np.random.choice's BY CONSTRUCTION (reference dla_gp.py:213-218).  The library's cdf and
    NumPy's differ by rounding, so uniforms planted exactly on cdf steps, one ulp below and one ulp above are where the
    two would disagree: the library counts every search that ends within 1e-11 of a step, and `settle_near_ties` redoes
    those levels with NumPy's arithmetic -- and, when an index changes, everything behind it.  Level by level: plant on
    the level's NumPy cdf, sweep, demand NumPy's indices.
    """
    from gpy_dla_detection_amd import parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch

    restore_options("level_spread", spread)
    rng = np.random.default_rng(11)
    sights = [make_sightline(80, n_unmasked=310, n_masked=6), make_sightline(81, n_unmasked=280, n_masked=0)]
    batch = DeviceBatch([device_sightline(s) for s in sights])
    S, max_dlas, nb = 3000, 4, 2
    zs = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    nhi = make_qmc_samples(sights[0], S)[2]
    min_sep = 3000.0 * 1000.0 / 299792458.0
    uniforms = parallel.resampling_uniforms(nb, max_dlas, S, seed=2)
    changed = 0
    for level in range(max_dlas - 1):
        # the sweep up to `level` does not depend on this level's uniforms: its likelihoods give the cdf to plant on
        ev, ll, base = batch.sweep(zs, nhi, max_dlas, min_sep, uniforms)
        for b in range(nb):
            uniforms[b, level] = planted_uniforms(numpy_cdf(ll[b, :, level]), rng)
        ev, ll, base = batch.sweep(zs, nhi, max_dlas, min_sep, uniforms)
        # three quarters of the searches sit on a step (behind a level whose indices changed the host has taken over,
        # and the count is a plain 'settle this level too')
        assert (batch.last_ties[:, level] > (S // 2 if level == 0 else 0)).all()
        changed += batch.last_changed
        for b in range(nb):
            for lower in range(level + 1):
                want = numpy_cdf(ll[b, :, lower]).searchsorted(uniforms[b, lower], side="right")
                np.testing.assert_array_equal(base[b, lower], want)
    assert changed > 0  # the device's own answer differed somewhere, and the levels behind it were redone
    # the redone levels are a sweep's levels: the oracle-driven host logic with the same indices injected gives the same numbers
    for b in range(nb):
        for level in range(1, max_dlas):
            idx = np.concatenate([np.arange(S)[None, :], base[b, :level]], axis=0)
            got = ll[b, :, level]
            too_close = np.any(np.diff(np.sort(zs[b][idx], axis=0), axis=0) < min_sep, axis=0)
            assert np.array_equal(np.isnan(got), too_close)
            direct = batch.sightlines[b].eval(zs[b][idx].T, nhi[idx].T) - np.log(S)
            np.testing.assert_allclose(got[~too_close], direct[~too_close], rtol=1e-12)
    # ordinary uniforms: a handful of near ties at most, nothing changes, and the by-sample split settles the same way
    uniforms = parallel.resampling_uniforms(nb, max_dlas, S, seed=3)
    ev, ll, base = batch.sweep(zs, nhi, max_dlas, min_sep, uniforms)
    assert batch.last_ties.sum() < 10
    batch.close()


def test_by_sample_split_settles_planted_ties_like_the_one_device_sweep():
    """the by-sample split (every rank finishes a level with the same uniforms) counts and settles near ties too"""
    import torch

    from gpy_dla_detection_amd import parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch

    torch.cuda.init()
    rng = np.random.default_rng(12)
    sight = make_sightline(82, n_unmasked=290, n_masked=4)
    batch = DeviceBatch([device_sightline(sight)])
    S, max_dlas = 2500, 3
    z = make_qmc_samples(sight, S)[0][None, :]
    nhi = make_qmc_samples(sight, S)[2]
    min_sep = 3000.0 * 1000.0 / 299792458.0
    uniforms = parallel.resampling_uniforms(1, max_dlas, S, seed=1)
    ev, ll, base = batch.sweep(z, nhi, max_dlas, min_sep, uniforms)
    uniforms[0, 0] = planted_uniforms(numpy_cdf(ll[0, :, 0]), rng)
    want = batch.sweep(z, nhi, max_dlas, min_sep, uniforms)
    np.testing.assert_array_equal(want[2][0, 0], numpy_cdf(want[1][0, :, 0]).searchsorted(uniforms[0, 0], side="right"))
    for slices in (None, [(0, 1000), (1000, 1003), (1003, S)]):
        got = parallel.sweep_sharded_samples_device(batch, z, nhi, max_dlas, min_sep, uniforms, slices=slices)
        np.testing.assert_array_equal(got[2], want[2])
        np.testing.assert_allclose(got[0], want[0], rtol=1e-12)
        np.testing.assert_allclose(got[1], want[1], rtol=1e-12, equal_nan=True)
    batch.close()


def test_device_sweep_stops_a_sightline_on_nan_evidence():
    """reference dla_gp.py:200-206: a NaN evidence ends that sightline's sweep; the others go on"""
    from gpy_dla_detection_amd import parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch

    sights = [make_sightline(i, n_unmasked=260, n_masked=3) for i in range(2)]
    batch = DeviceBatch([device_sightline(s) for s in sights])
    S = 96
    zs = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    nhi = np.stack([make_qmc_samples(sights[0], S)[2]] * 2)   # per-sightline column densities (nhi_stride = S)
    nhi[1, :] = np.nan                                         # every sample of sightline 1 evaluates to NaN
    min_sep = 3000.0 * 1000.0 / 299792458.0
    ev, ll, base = parallel.sweep_batch_device(batch, zs, nhi, 3, min_sep, seed=0)
    assert np.isnan(ev[1]).all() and np.isnan(ll[1]).all() and not base[1].any()
    alone = parallel.sweep_batch_device(DeviceBatch([device_sightline(sights[0])]), zs[:1], nhi[0], 3, min_sep, seed=0)
    np.testing.assert_array_equal(base[0], alone[2][0])
    np.testing.assert_allclose(ev[0], alone[0][0], rtol=1e-12)
    assert np.isfinite(ev[0]).all()
    with pytest.raises(ValueError):
        batch.sweep(zs, nhi[0], 9, min_sep, np.zeros((2, 8, S)))     # max_dlas > DLAGP_MAX_ABSORBERS
    with pytest.raises(ValueError):
        batch.sweep(zs, nhi[0], 2, min_sep)                          # uniforms missing


# ------------------------------------------------------------------------------------------------
# f1: NullGP.get_interp on the device (dlagp_model_create / dlagp_create_from_model)
# ------------------------------------------------------------------------------------------------
def test_device_interpolation_matches_reference_golden():
    """model interpolation, mean-flux suppression and noise scaling on the device vs reference set_data output"""
    from gpy_dla_detection_amd.null_gp import NullGP
    from gpy_dla_detection_amd.set_parameters import Parameters
    from gpy_dla_detection_amd.sightline import DeviceModel, DeviceSightline
    from tests.golden_inputs import raw_spectrum_and_model

    g = load_golden("set_data.npz")
    z_qso, rest, flux, nv, mask, rest_grid, mu, M, log_omega, lc0, lt0, lb = raw_spectrum_and_model()
    params = Parameters()
    host = NullGP(params, FixedPrior(), rest_grid, mu, M, log_omega, lc0, lt0, lb)
    host.set_data(rest, flux, nv, mask, z_qso, normalize=True, build_model=False)   # index logic only
    model = DeviceModel(rest_grid, mu, np.asfortranarray(M), log_omega, lc0, lt0, lb,
                        num_forest_lines=params.num_forest_lines)
    dev = DeviceSightline.from_model(model, host.x, host.y, host.v, host.this_wavelengths, z_qso,
                                     host.padded_wavelengths, ~host.pixel_mask[host.ind_unmasked])
    this_mu, this_M, this_omega2 = dev.read_model()
    np.testing.assert_allclose(this_mu, g["this_mu"], rtol=1e-14)
    np.testing.assert_allclose(this_M, g["this_M"], rtol=1e-14, atol=1e-16)  # atol: interpolants crossing zero
    np.testing.assert_allclose(this_omega2, g["this_omega2"], rtol=1e-14)
    assert abs(dev.null() - g["null_evidence"]) <= RTOL_LL * abs(g["null_evidence"])
    with pytest.raises(ValueError):   # scipy's interp1d refuses to extrapolate
        DeviceSightline.from_model(model, host.x + 400.0, host.y, host.v, host.this_wavelengths, z_qso,
                                   host.padded_wavelengths, ~host.pixel_mask[host.ind_unmasked])


def test_deferred_context_set_up_completes_with_the_batch():
    """
    DLAGP_FLAG_DEFER_SYNC (the catalogue driver): contexts created without waiting for their uploads and set-up kernels are
    complete once the same thread has built a batch with them -- same numbers as contexts created one wait at a time -- and
    a pending context may be destroyed without ever having been waited for.
    """
    from gpy_dla_detection_amd.null_gp import NullGP
    from gpy_dla_detection_amd.set_parameters import Parameters
    from gpy_dla_detection_amd.sightline import DeviceBatch, DeviceModel, DeviceSightline, setup_sync
    from tests.golden_inputs import raw_spectrum_and_model

    z_qso, rest, flux, nv, mask, rest_grid, mu, M, log_omega, lc0, lt0, lb = raw_spectrum_and_model()
    params = Parameters()
    host = NullGP(params, FixedPrior(), rest_grid, mu, M, log_omega, lc0, lt0, lb)
    host.set_data(rest, flux, nv, mask, z_qso, normalize=True, build_model=False)
    model = DeviceModel(rest_grid, mu, np.asfortranarray(M), log_omega, lc0, lt0, lb, num_forest_lines=params.num_forest_lines)
    keep = ~host.pixel_mask[host.ind_unmasked]

    def make(defer, scale=1.0):
        return DeviceSightline.from_model(model, host.x, scale * host.y, host.v, host.this_wavelengths, z_qso,
                                          host.padded_wavelengths, keep, defer_sync=defer)

    S = 200
    z = 2.2 + 0.5 * halton(S, 2)
    nhi = 10.0 ** (20.0 + 2.5 * halton(S, 3))
    waited = DeviceBatch([make(False, 1.0 + 0.01 * i) for i in range(6)])
    want, want_null = waited.eval(np.tile(z, (6, 1)), np.tile(nhi, (6, 1))), waited.nulls()
    deferred = [make(True, 1.0 + 0.01 * i) for i in range(6)]
    batch = DeviceBatch(deferred)  # ends with the one synchronisation
    np.testing.assert_array_equal(batch.eval(np.tile(z, (6, 1)), np.tile(nhi, (6, 1))), want)
    np.testing.assert_array_equal(batch.nulls(), want_null)
    single = make(True)
    setup_sync(0)
    assert abs(single.null() - want_null[0]) <= 1e-13 * abs(want_null[0])
    for _ in range(20):  # never waited for: destroy must cope
        make(True).close()
    assert abs(make(False).null() - want_null[0]) <= 1e-13 * abs(want_null[0])


def test_deferred_contexts_may_cross_threads():
    """
    The deferred set-up is self-enforcing: every context carries an event recorded behind its set-up work on the
    creating thread's stream, and its first use waits for it -- a batch built on ANOTHER thread, or a plain evaluation
    with no synchronisation call in between -- so the numbers are those of contexts created one wait at a time.
    """
    import threading

    from gpy_dla_detection_amd.null_gp import NullGP
    from gpy_dla_detection_amd.set_parameters import Parameters
    from gpy_dla_detection_amd.sightline import DeviceBatch, DeviceModel, DeviceSightline
    from tests.golden_inputs import raw_spectrum_and_model

    z_qso, rest, flux, nv, mask, rest_grid, mu, M, log_omega, lc0, lt0, lb = raw_spectrum_and_model()
    params = Parameters()
    host = NullGP(params, FixedPrior(), rest_grid, mu, M, log_omega, lc0, lt0, lb)
    host.set_data(rest, flux, nv, mask, z_qso, normalize=True, build_model=False)
    model = DeviceModel(rest_grid, mu, np.asfortranarray(M), log_omega, lc0, lt0, lb, num_forest_lines=params.num_forest_lines)
    keep = ~host.pixel_mask[host.ind_unmasked]

    def make(defer, scale):
        return DeviceSightline.from_model(model, host.x, scale * host.y, host.v, host.this_wavelengths, z_qso,
                                          host.padded_wavelengths, keep, defer_sync=defer)

    S, nb = 200, 12
    z = np.tile(2.2 + 0.5 * halton(S, 2), (nb, 1))
    nhi = np.tile(10.0 ** (20.0 + 2.5 * halton(S, 3)), (nb, 1))
    waited = DeviceBatch([make(False, 1.0 + 0.01 * i) for i in range(nb)])
    want, want_null = waited.eval(z, nhi), waited.nulls()
    for _ in range(5):
        made, got = [], {}

        def creator():
            made.extend(make(True, 1.0 + 0.01 * i) for i in range(nb))

        a = threading.Thread(target=creator)
        a.start()
        a.join()  # thread A is gone; nobody synchronised its stream

        def user():
            batch = DeviceBatch(made)
            got["ll"], got["null"] = batch.eval(z, nhi), batch.nulls()
            batch.close()

        b = threading.Thread(target=user)
        b.start()
        b.join()
        np.testing.assert_array_equal(got["ll"], want)
        np.testing.assert_array_equal(got["null"], want_null)
        for dev in made:
            dev.close()
    # a deferred context used directly: its first evaluation waits for its event
    for i in range(10):
        single = make(True, 1.0 + 0.01 * (i % nb))
        assert single.null() == want_null[i % nb]
        single.close()
    mu_back = make(True, 1.0).read_model()[0]
    np.testing.assert_array_equal(mu_back, make(False, 1.0).read_model()[0])


def test_negative_noise_variance_is_nan_or_linalgerror_like_the_reference():
    """
    Invalid input, reference semantics: with a negative noise variance d = omega2 a^2 + v is negative at some pixels
    and `np.log(d)` (null_gp.py:356) makes the density NaN -- for ANY number of negative pixels -- unless B stops being
    positive definite first (LinAlgError, null_gp.py:343).  A running product of d (the MFMA kernel) would turn an even
    number of negative pixels into a finite number, so such a sightline runs on the generic kernel.
    """
    import warnings

    from gpy_dla_detection_amd.sightline import DeviceBatch, DeviceSightline, get_option, log_mvnpdf_low_rank
    from oracle import ref_like as R

    sight = make_sightline(3, n_unmasked=900, n_masked=20)
    z, _, nhi = make_qmc_samples(sight, 96)
    for count in (1, 2, 4):  # odd and even numbers of negative pixels
        v = sight.v.copy()
        pixels = np.arange(count) * 37 + 100
        v[pixels] = -(sight.this_omega2[pixels] + 1.0)  # d = -1 there for the null model, B stays positive definite
        dev = DeviceSightline(sight.y, sight.this_mu, sight.this_M, sight.this_omega2, v, sight.padded_wavelengths, sight.keep)
        st = R.SpectrumState(sight.y, v, sight.this_mu, sight.this_M, sight.this_omega2, sight.padded_wavelengths,
                             sight.unmasked_wavelengths, sight.keep)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want_null = R.null_log_model_evidence(st)
            want = np.array([R.sample_log_likelihood_k_dlas(st, [z[i]], [nhi[i]]) for i in range(8)])
        assert np.isnan(want_null) and np.all(np.isnan(want))  # the reference's answer
        assert np.isnan(dev.null())
        got = dev.eval(z, nhi)
        assert get_option("last_launch_path") == 1 and np.all(np.isnan(got))
        clean = DeviceSightline.from_state(sight)
        both = DeviceBatch([clean, dev]).eval(np.stack([z, z]), np.stack([nhi, nhi]))
        np.testing.assert_allclose(both[0], clean.eval(z, nhi), rtol=1e-9)  # the valid member is untouched
        assert np.all(np.isnan(both[1]))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert np.isnan(log_mvnpdf_low_rank(sight.y, sight.this_mu, sight.this_M, sight.this_omega2 + v))
    # a negative variance large enough to make B indefinite: LinAlgError, as from np.linalg.cholesky
    neg = DeviceSightline(sight.y, sight.this_mu, 10.0 * sight.this_M, sight.this_omega2, -np.abs(sight.v),
                          sight.padded_wavelengths, sight.keep)
    with pytest.raises(np.linalg.LinAlgError):
        neg.eval(z, nhi)
    assert get_option("last_launch_path") == 1


# ------------------------------------------------------------------------------------------------
# f3: the catalogue driver end to end on the device vs the reference's own loop body (G8)
# ------------------------------------------------------------------------------------------------
def test_process_qso_on_device_matches_reference_driver_golden():
    from gpy_dla_detection_amd.run_bayes_select import process_qso
    from tests.test_host_logic import FixedPrior as Prior, assert_catalogue_matches_golden, catalogue_inputs

    g = load_golden("catalogue.npz")
    learned, spectra, z_qsos, params, dla_samples, sub_samples = catalogue_inputs(int(g["S"]))
    results = process_qso(spectra, z_qsos, params, Prior(), dla_samples, sub_samples, learned,
                          max_dlas=int(g["max_dlas"]), batch_size=2)
    assert_catalogue_matches_golden(results, g, rtol=RTOL_LL)


def test_native_batch_preparation_equals_the_per_spectrum_python():
    """
    f1, the data half: dlagp_batch_prepare -- normalisation, cuts, masks, padded grid, redshift bounds, sample tables,
    contexts and batch of a whole batch of RAW spectra in one call (reference null_gp.py:117-177,
    set_parameters.py:125-159, dla_samples.py:94-104) -- against `prepare_data` + `Parameters.z_dla_range` +
    `DeviceSightline.from_model` + `DeviceBatch` per spectrum in Python: the contexts hold the same bits, the tables are
    the same bits, and a catalogue run gives identical output arrays either way.
    """
    from gpy_dla_detection_amd.null_gp import prepare_data
    from gpy_dla_detection_amd.run_bayes_select import DeviceEngine, process_qso
    from gpy_dla_detection_amd.samples import DLASamples, SubDLASamples
    from gpy_dla_detection_amd.set_parameters import Parameters
    from gpy_dla_detection_amd.sightline import SpectrumTable
    from gpy_dla_detection_amd.synthetic import make_catalogue
    from tests.test_host_logic import FixedPrior

    S, max_dlas, num = 640, 3, 44
    params = Parameters(num_dla_samples=S)
    u = halton(S, 3)
    dla = DLASamples(params, None, halton(S, 2), 20.0 + 3.0 * u)
    sub_same = SubDLASamples(params, None, halton(S, 2), 19.5 + 0.5 * u, Z_dla=1.0, Z_lls=0.25)
    sub_other = SubDLASamples(params, None, halton(S, 5), 19.5 + 0.5 * u, Z_dla=1.0, Z_lls=0.25)
    learned, spectra, z_qsos = make_catalogue(num, mix="dr12q")
    # (1) one batch, looked at closely
    engine = DeviceEngine(learned, params)
    table = SpectrumTable(spectra, z_qsos)
    assert table.ok
    rows = np.array([3, 0, 17, 41, 8])
    batch, sightlines, z_dla, z_lls, scalars = engine.prepare_batch(table, rows, True, dla.offset_samples, sub_other.offset_samples)
    for i, q in enumerate(rows):
        wavelengths, flux, noise_variance, mask = spectra[q]
        st = prepare_data(params, params.emitted_wavelengths(wavelengths, z_qsos[q]), flux, noise_variance, mask, z_qsos[q])
        reference = engine.sightline(st, z_qsos[q], True)
        for got, want in zip(sightlines[i].read_model(), reference.read_model()):
            np.testing.assert_array_equal(got, want)
        assert sightlines[i].n == reference.n == int(scalars[i, 5])
        np.testing.assert_array_equal(z_dla[i], dla.sample_z_dlas(st["this_wavelengths"], z_qsos[q]))
        np.testing.assert_array_equal(z_lls[i], sub_other.sample_z_lls(st["this_wavelengths"], z_qsos[q]))
        assert (scalars[i, 1], scalars[i, 2]) == params.z_dla_range(st["this_wavelengths"], z_qsos[q])
        assert (scalars[i, 3], scalars[i, 4]) == params.z_dla_range(wavelengths, z_qsos[q])
        assert scalars[i, 0] == st["normalization_median"]
        assert sightlines[i].null() == reference.null()  # y, v and the padded grid are in there
        zz = dla.sample_z_dlas(st["this_wavelengths"], z_qsos[q])[:64]
        np.testing.assert_array_equal(sightlines[i].eval(zz, dla.nhi_samples[:64]), reference.eval(zz, dla.nhi_samples[:64]))
        reference.close()
    engine.release(batch, sightlines)
    # a spectrum the Python raises on (no pixel inside the modelled window) fails the call and leaves nothing behind
    bad = list(spectra[:3]) + [(spectra[0][0][:800], spectra[0][1][:800], spectra[0][2][:800], spectra[0][3][:800])]
    with pytest.raises(ValueError, match="modelled window"):
        engine.prepare_batch(SpectrumTable(bad, [2.5, 2.6, 2.7, 9.0]), np.arange(4), True, dla.offset_samples)
    engine.close()
    # (2) whole catalogue runs: every output array identical, pipelined and in order, shared and separate offsets
    for sub in (sub_same, sub_other):
        for prefetch in (False, True):
            native = process_qso(spectra, z_qsos, params, FixedPrior(), dla, sub, learned, max_dlas=max_dlas, batch_size=8,
                                 prefetch=prefetch)
            python = process_qso(spectra, z_qsos, params, FixedPrior(), dla, sub, learned, max_dlas=max_dlas, batch_size=8,
                                 prefetch=prefetch, native_prepare=False)
            assert sorted(native) == sorted(python)
            for name in native:
                np.testing.assert_array_equal(native[name], python[name], err_msg=name)
    # spectra that are not float64 keep the Python path (its arithmetic runs in their type)
    odd = [(w, f.astype(np.float32), v, m) for w, f, v, m in spectra[:6]]
    assert not SpectrumTable(odd, z_qsos[:6]).ok
    process_qso(odd, z_qsos[:6], params, FixedPrior(), dla, sub_same, learned, max_dlas=2, batch_size=3, prefetch=False)


def test_pipelined_driver_equals_the_in_order_driver_on_the_device():
    """
    42 spectra in batches of 8: the pipelined driver (five threads, short first batches, one table of sample redshifts
    shared by both models, the batch's bookkeeping as array expressions) against the same catalogue swept in order on
    the calling thread, and -- for a few spectra -- against the drop-in classes called one spectrum at a time the way the
    reference's driver does (run_bayes_select.py:141-230).  Different batch compositions, different launch shapes: the
    log-likelihoods agree to rounding, everything discrete exactly.
    """
    from gpy_dla_detection_amd.bayesian_model_selection import BayesModelSelect
    from gpy_dla_detection_amd.dla_gp import DLAGP
    from gpy_dla_detection_amd.null_gp import NullGP
    from gpy_dla_detection_amd.run_bayes_select import process_qso
    from gpy_dla_detection_amd.subdla_gp import SubDLAGP
    from tests.test_host_logic import FixedPrior, catalogue_inputs

    S, max_dlas = 512, 3
    learned, spectra, z_qsos, params, dla_samples, sub_samples = catalogue_inputs(S)
    rng = np.random.default_rng(17)
    many, many_z = [], []
    for i in range(42):  # the three golden spectra with fresh noise and slightly different redshifts
        w, f, v, m = spectra[i % 3]
        many.append((w, f + 0.02 * rng.standard_normal(f.shape), v, m))
        many_z.append(z_qsos[i % 3] + 0.002 * (i // 3))
    piped = process_qso(many, many_z, params, FixedPrior(), dla_samples, sub_samples, learned, max_dlas=max_dlas, batch_size=8)
    in_order = process_qso(many, many_z, params, FixedPrior(), dla_samples, sub_samples, learned, max_dlas=max_dlas,
                           batch_size=8, prefetch=False)
    exact = ("base_sample_inds", "min_z_dlas", "max_z_dlas", "log_priors_no_dla", "log_priors_lls", "log_priors_dla", "z_qsos")
    for name in piped:
        if name in exact:
            np.testing.assert_array_equal(piped[name], in_order[name], err_msg=name)
        elif piped[name].dtype.kind == "f" and piped[name].ndim:
            assert np.array_equal(np.isnan(piped[name]), np.isnan(in_order[name])), name
            np.testing.assert_allclose(piped[name], in_order[name], rtol=1e-9, atol=1e-280, equal_nan=True, err_msg=name)
    assert np.array_equal(np.argmax(piped["model_posteriors"], axis=1), np.argmax(in_order["model_posteriors"], axis=1))

    bayes = BayesModelSelect([0, 1, max_dlas], 2)
    model = tuple(learned)[:7]
    for q in (0, 13, 41):
        np.random.seed(0)
        w, f, v, m = many[q]
        rest = params.emitted_wavelengths(w, many_z[q])
        gps = [NullGP(params, FixedPrior(), *model),
               SubDLAGP(params, FixedPrior(), sub_samples, *model, min_z_separation=3000.0, broadening=True),
               DLAGP(params, FixedPrior(), dla_samples, *model, min_z_separation=3000.0, broadening=True)]
        for gp in gps:
            gp.set_data(rest, f, v, m, many_z[q], build_model=True)
        bayes.model_selection(gps, many_z[q])
        np.testing.assert_allclose(piped["log_posteriors_dla"][q], bayes.log_posteriors[-max_dlas:], rtol=1e-9)
        np.testing.assert_allclose(piped["log_posteriors_no_dla"][q], bayes.log_posteriors[0], rtol=1e-9)
        np.testing.assert_allclose(piped["log_posteriors_lls"][q], bayes.log_posteriors[1], rtol=1e-9)
        np.testing.assert_array_equal(piped["base_sample_inds"][q], gps[2].base_sample_inds.T)
        np.testing.assert_allclose(piped["sample_log_likelihoods_dla"][q], gps[2].sample_log_likelihoods, rtol=1e-9, equal_nan=True)
        np.testing.assert_allclose(piped["model_posteriors"][q], bayes.model_posteriors, rtol=1e-6, atol=1e-280)


# ------------------------------------------------------------------------------------------------
# INTEGRATION route A: the reference driver's loop body (run_bayes_select.py:141-230) with the imports swapped
# ------------------------------------------------------------------------------------------------
def test_reference_driver_loop_with_swapped_imports_matches_golden():
    """
    What a maintainer gets by changing only the import lines of the reference's run_bayes_select.py: three
    GP objects per spectrum, `set_data` on each, `BayesModelSelect.model_selection`, `maximum_a_posteriori`,
    `np.random.seed(0)` per spectrum -- against every array of golden set G8 (the reference's own classes
    in the same loop).
    """
    from gpy_dla_detection_amd.bayesian_model_selection import BayesModelSelect
    from gpy_dla_detection_amd.dla_gp import DLAGP
    from gpy_dla_detection_amd.null_gp import NullGP
    from gpy_dla_detection_amd.subdla_gp import SubDLAGP
    from tests.test_host_logic import FixedPrior, assert_catalogue_matches_golden, catalogue_inputs

    g = load_golden("catalogue.npz")
    S, max_dlas = int(g["S"]), int(g["max_dlas"])
    learned, spectra, z_qso_list, param, dla_samples, subdla_samples = catalogue_inputs(S)
    learned = tuple(learned)[:7]
    prior = FixedPrior()
    broadening = True
    from tests.golden_inputs import run_driver_loop

    out = run_driver_loop(
        spectra, z_qso_list, param,
        lambda: (NullGP(param, prior, *learned),
                 DLAGP(param, prior, dla_samples, *learned, min_z_separation=3000.0, broadening=broadening),
                 SubDLAGP(param, prior, subdla_samples, *learned, min_z_separation=3000.0, broadening=broadening)),
        BayesModelSelect([0, 1, max_dlas], 2), max_dlas)
    assert out["base_sample_inds"].dtype == np.int32 and out["sample_log_likelihoods_dla"].shape == (len(spectra), S, max_dlas)
    out["z_qsos"] = np.array(z_qso_list)
    out["num_dla_samples"], out["k"] = param.num_dla_samples, param.k
    assert_catalogue_matches_golden(out, g, rtol=RTOL_LL)


def route_a_models(spectrum, z_qso, S, learned=None, device=0):
    """the reference driver's three GP objects for one spectrum (run_bayes_select.py:151-186), imports swapped"""
    from gpy_dla_detection_amd.dla_gp import DLAGP
    from gpy_dla_detection_amd.null_gp import NullGP
    from gpy_dla_detection_amd.subdla_gp import SubDLAGP
    from tests.test_host_logic import FixedPrior, catalogue_inputs

    stock, _, _, param, dla_samples, subdla_samples = catalogue_inputs(S)
    learned = tuple(stock)[:7] if learned is None else learned
    prior = FixedPrior()
    wavelengths, flux, noise_variance, pixel_mask = spectrum
    rest = param.emitted_wavelengths(wavelengths, z_qso)
    gp = NullGP(param, prior, *learned)
    dla_gp = DLAGP(param, prior, dla_samples, *learned, min_z_separation=3000.0, broadening=True)
    subdla_gp = SubDLAGP(param, prior, subdla_samples, *learned, min_z_separation=3000.0, broadening=True)
    for model in (gp, dla_gp, subdla_gp):
        model.set_data(rest, flux, noise_variance, pixel_mask, z_qso, build_model=True)
    return gp, subdla_gp, dla_gp


def test_the_three_gp_objects_of_a_spectrum_share_one_sightline_and_one_call():
    """
    Route A's per-spectrum cost: the three GP objects find ONE HBM-resident sightline (built on the device from the
    learned model), and `BayesModelSelect.model_selection` sweeps the three models with one call into the library --
    the same numbers, side effects and RNG state, bit for bit, as the model-by-model loop, and as three objects that
    do not share (the model half read on the host first).
    """
    from gpy_dla_detection_amd import sightline as sl
    from gpy_dla_detection_amd.bayesian_model_selection import BayesModelSelect
    from tests.test_host_logic import catalogue_inputs

    S, max_dlas = 512, 4
    _, spectra, z_qsos, _, _, _ = catalogue_inputs(S)
    calls = {"model_evidences": 0, "sweep": 0}
    real_me, real_sweep = sl.DeviceBatch.model_evidences_finish, sl.DeviceBatch.sweep

    def counted(name, fn):
        def wrapper(self, *a, **k):
            calls[name] += 1
            return fn(self, *a, **k)
        return wrapper

    sl.DeviceBatch.model_evidences_finish = counted("model_evidences", real_me)  # the two-phase form of the one call
    sl.DeviceBatch.sweep = counted("sweep", real_sweep)
    try:
        for spectrum, z_qso in zip(spectra, z_qsos):
            models = route_a_models(spectrum, z_qso, S)
            np.random.seed(0)
            fused = BayesModelSelect([0, 1, max_dlas], 2)
            fused.model_selection(list(models), z_qso)
            state_fused = np.random.get_state()[1].copy()
            assert models[0].device_sightline() is models[1].device_sightline() is models[2].device_sightline()
            assert models[0]._shared is models[2]._shared and models[0]._shared.batch is models[1]._shared.batch
            ll_fused = (models[1].sample_log_likelihoods.copy(), models[2].sample_log_likelihoods.copy(),
                        models[2].base_sample_inds.copy())
            assert ll_fused[1].shape == (S, max_dlas) and ll_fused[2].shape == (max_dlas - 1, S)
            # the plain loop of the reference's model_selection on the same objects
            np.random.seed(0)
            loop = [models[0].log_model_evidence()] + list(models[1].log_model_evidences(1)) + \
                list(models[2].log_model_evidences(max_dlas))
            np.testing.assert_array_equal(fused.log_likelihoods, np.array(loop))
            np.testing.assert_array_equal(np.random.get_state()[1], state_fused)
            np.testing.assert_array_equal(models[1].sample_log_likelihoods, ll_fused[0])
            np.testing.assert_array_equal(models[2].sample_log_likelihoods, ll_fused[1])
            np.testing.assert_array_equal(models[2].base_sample_inds, ll_fused[2])
            # objects whose model half was read on the host first: private sightlines from the host arrays, no fused call
            private = route_a_models(spectrum, z_qso, S)
            for m in private:
                assert m.this_M.shape[1] == 20
            np.random.seed(0)
            apart = BayesModelSelect([0, 1, max_dlas], 2)
            apart.model_selection(list(private), z_qso)
            assert private[0].device_sightline() is not private[2].device_sightline() and private[0]._shared is None
            np.testing.assert_allclose(apart.log_likelihoods, fused.log_likelihoods, rtol=1e-11)
            np.testing.assert_array_equal(private[2].base_sample_inds, ll_fused[2])
    finally:
        sl.DeviceBatch.model_evidences_finish, sl.DeviceBatch.sweep = real_me, real_sweep
    assert calls["model_evidences"] == len(spectra) and calls["sweep"] == 4 * len(spectra)


def test_in_place_edits_reach_the_device():
    """
    The reference reads live arrays at every call.  Here the sightline lives in HBM, so what the sweep reads is
    compared by content on every call: an in-place edit of the spectrum (`gp.y[:] = ...`), of the model half, or of
    the learned arrays between two constructions is evaluated, never a stale copy.
    """
    from tests.test_host_logic import catalogue_inputs

    S = 256
    learned, spectra, z_qsos, _, _, _ = catalogue_inputs(S)
    learned = tuple(learned)[:7]
    gp, sub, dla = route_a_models(spectra[0], z_qsos[0], S)
    before = gp.log_model_evidence()
    shared = gp._shared
    assert shared is not None and dla.device_sightline() is gp.device_sightline()
    gp.y[:] = 0.9 * gp.y  # in place: same array object
    after = gp.log_model_evidence()
    assert after != before and gp._shared is not shared
    assert dla._shared is None or dla.device_sightline() is not gp.device_sightline()  # dla's own y is untouched
    fresh, _, _ = route_a_models(spectra[0], z_qsos[0], S)
    fresh.y = 0.9 * fresh.y  # the same data as a new array
    assert fresh.log_model_evidence() == after
    # the model half on the host: edits in place are seen too
    mu = gp.this_mu  # materialised on the host now -> private sightline from these arrays
    host = gp.log_model_evidence()
    np.testing.assert_allclose(host, after, rtol=1e-12)
    mu *= 1.01
    edited = gp.log_model_evidence()
    assert edited != host
    twin, _, _ = route_a_models(spectra[0], z_qsos[0], S)
    twin.y = 0.9 * twin.y
    twin.this_mu = 1.01 * twin.this_mu
    assert twin.log_model_evidence() == edited
    # the learned factor edited in place between two constructions: the second object is another model on the device
    rest_grid, mu_l, M_l, log_omega = (np.array(a, copy=True) for a in learned[:4])
    mine = (rest_grid, mu_l, M_l, log_omega) + learned[4:]
    first = route_a_models(spectra[1], z_qsos[1], S, learned=mine)[2]
    ev_first = first.log_model_evidences(2)
    M_l[[100, 200]] = M_l[[200, 100]]  # sum-preserving, same array object
    M_l[:, 3] *= 1.5
    second = route_a_models(spectra[1], z_qsos[1], S, learned=mine)[2]
    np.random.seed(0)
    ev_second = second.log_model_evidences(2)
    assert not np.array_equal(ev_first, ev_second)
    copy = route_a_models(spectra[1], z_qsos[1], S, learned=tuple(np.array(a, copy=True) for a in mine[:4]) + mine[4:])[2]
    np.random.seed(0)
    np.testing.assert_array_equal(copy.log_model_evidences(2), ev_second)


def test_overridden_this_dla_gp_is_honoured_sample_by_sample():
    """
    The hook of reference examples/gp_find_lls.py:159-224: a subclass that overrides `this_dla_gp` changes
    the model, so the sweep must call it for every sample (no batched launch of the stock profile) and
    agree with the oracle evaluated on the modified model.
    """
    from gpy_dla_detection_amd.dla_gp import DLAGP
    from gpy_dla_detection_amd.set_parameters import Parameters
    from oracle import ref_like as R
    from tests.golden_inputs import raw_spectrum_and_model

    calls = []

    class DimmedDLAGP(DLAGP):
        def this_dla_gp(self, z_dlas, nhis):
            calls.append(len(z_dlas))
            mu, M, omega2 = super().this_dla_gp(z_dlas, nhis)
            dim = 1.0 - 0.1 * np.exp(-0.5 * ((self.this_wavelengths - 4300.0) / 150.0) ** 2)  # an extra, fixed factor
            return mu * dim, M * dim[:, None], omega2 * dim ** 2

    S, max_dlas = 48, 2
    z_qso, rest, flux, noise_variance, pixel_mask, rest_grid, mu, M, log_omega, lc0, lt0, lb = raw_spectrum_and_model()
    params = Parameters(num_dla_samples=S)
    from gpy_dla_detection_amd.samples import DLASamples

    samples = DLASamples(params, None, halton(S, 2), 20.0 + 3.0 * halton(S, 3))
    gp = DimmedDLAGP(params, None, samples, rest_grid, mu, M, log_omega, lc0, lt0, lb)
    gp.set_data(rest, flux, noise_variance, pixel_mask, z_qso, build_model=True)
    np.random.seed(0)
    got = gp.log_model_evidences(max_dlas)
    assert len(calls) == S * max_dlas and calls[0] == 1 and calls[-1] == 2  # the per-sample path was taken

    # oracle on the same modified model, the reference's loop (dla_gp.py:92-225) with the same seed
    st = R.SpectrumState(gp.y, gp.v, gp.this_mu, gp.this_M, gp.this_omega2, gp.padded_wavelengths,
                         gp.unmasked_wavelengths, ~gp.pixel_mask[gp.ind_unmasked])
    dim = 1.0 - 0.1 * np.exp(-0.5 * ((gp.this_wavelengths - 4300.0) / 150.0) ** 2)

    def sample_ll(z_dlas, nhis):
        a = np.ones_like(st.this_mu)
        for z, n in zip(z_dlas, nhis):
            a = a * R.voigt_absorption(st.padded_wavelengths, n, z)[st.keep]
        a = a * dim
        return R.log_mvnpdf_low_rank(st.y, st.this_mu * a, st.this_M * a[:, None], st.this_omega2 * a ** 2 + st.v)

    from gpy_dla_detection_amd.dla_gp import occupation_sweep

    np.random.seed(0)
    sample_z = samples.sample_z_dlas(gp.this_wavelengths, z_qso)
    want, want_ll, want_base = occupation_sweep(
        lambda z, nhi: np.array([sample_ll(z[i], nhi[i]) for i in range(z.shape[0])]),
        sample_z, samples.nhi_samples, max_dlas, gp.min_z_separation, S)
    np.testing.assert_allclose(got, want, rtol=RTOL_LL)
    np.testing.assert_allclose(gp.sample_log_likelihoods, want_ll, rtol=RTOL_LL, equal_nan=True)
    assert np.array_equal(gp.base_sample_inds, want_base)
    # a single overridden sample, too
    one = gp.sample_log_likelihood_k_dlas(np.array([sample_z[3]]), np.array([samples.nhi_samples[3]]))
    assert abs(one - sample_ll([sample_z[3]], [samples.nhi_samples[3]])) <= RTOL_LL * abs(one)


# ------------------------------------------------------------------------------------------------
# f4: MCMC target density (log_posterior_mcmc) on the device
# ------------------------------------------------------------------------------------------------
def test_log_posterior_mcmc_matches_reference_golden():
    from gpy_dla_detection_amd import log_posterior_mcmc as mc
    from tests.golden_inputs import mcmc_pdf, mcmc_thetas

    g = load_golden("log_posterior.npz")
    sight = make_sightline(1, n_unmasked=400, n_masked=9)
    assert checksum(sight.y, sight.this_M) == int(g["sight_crc"])
    z_lo, z_hi = sight.z_range()
    thetas = mcmc_thetas(z_lo, z_hi)
    np.testing.assert_array_equal(thetas, g["thetas"])
    pixel_mask = ~sight.keep
    ind_unmasked = np.ones(sight.keep.shape[0], dtype=bool)
    args = (sight.this_wavelengths, sight.y, sight.v, sight.z_qso, z_lo, z_hi, 20.0, 23.0, mcmc_pdf,
            sight.padded_wavelengths, sight.this_mu, sight.this_M, sight.this_omega2, pixel_mask, ind_unmasked, 3)
    batched = mc.log_posterior_vectorized(thetas, *args)          # emcee vectorize=True form: one launch
    assert np.array_equal(np.isneginf(batched), np.isneginf(g["log_posterior"]))
    live = np.isfinite(g["log_posterior"])
    np.testing.assert_allclose(batched[live], g["log_posterior"][live], rtol=RTOL_LL)
    single = np.array([mc.log_posterior(tuple(t), *args) for t in thetas[:3]])   # the reference's call form
    np.testing.assert_allclose(single, batched[:3], rtol=1e-13)
    # two absorbers per walker: (z_1, z_2, logN_1, logN_2) -- against the sightline API
    pair = np.concatenate([thetas[:2, :1].T, thetas[:2, 1:].T], axis=1)
    want = device_sightline(sight).eval(pair[:, :2], 10.0 ** pair[:, 2:])[0] + np.log(mcmc_pdf(pair[0, 2])) + \
        np.log(mcmc_pdf(pair[0, 3]))
    assert abs(mc.log_posterior_vectorized(pair, *args)[0] - want) <= 1e-12 * abs(want)


def test_log_posterior_mcmc_sees_in_place_edits_like_the_reference():
    """
    The reference's `log_posterior` re-reads its arrays on every call (log_posterior_mcmc.py:139-250): editing `y` in
    place between two calls changes the answer.  The device route keeps one HBM copy of the arrays and keys it by their
    CONTENT (round 5; by identity before: the edit went unseen): same arrays object, new bytes -> new sightline, the
    oracle's value on the edited flux; the edit undone -> the first value again, bit for bit; calls from several
    threads at once (an emcee pool) share the cache under its lock.
    """
    import threading

    from gpy_dla_detection_amd import log_posterior_mcmc as mc
    from oracle import ref_like as R
    from tests.golden_inputs import mcmc_pdf, mcmc_thetas

    sight = make_sightline(3, n_unmasked=350, n_masked=7)
    z_lo, z_hi = sight.z_range()
    thetas = mcmc_thetas(z_lo, z_hi)
    live = [t for t in thetas if z_lo < t[0] < z_hi and 20.0 < t[1] < 23.0][:3]
    assert live
    pixel_mask = ~sight.keep
    ind_unmasked = np.ones(sight.keep.shape[0], dtype=bool)
    y = sight.y.copy()
    args = (sight.this_wavelengths, y, sight.v, sight.z_qso, z_lo, z_hi, 20.0, 23.0, mcmc_pdf,
            sight.padded_wavelengths, sight.this_mu, sight.this_M, sight.this_omega2, pixel_mask, ind_unmasked, 3)

    def oracle_value(t):
        st = R.SpectrumState(y, sight.v, sight.this_mu, sight.this_M, sight.this_omega2, sight.padded_wavelengths,
                             sight.unmasked_wavelengths, sight.keep)
        return R.sample_log_likelihood_k_dlas(st, [t[0]], [10.0 ** t[1]]) + np.log(mcmc_pdf(t[1]))

    t = tuple(live[0])
    before = mc.log_posterior(t, *args)
    assert abs(before - oracle_value(t)) <= RTOL_LL * abs(before)
    y[40:60] *= 0.5  # in place: the same array object
    after = mc.log_posterior(t, *args)
    assert after != before
    assert abs(after - oracle_value(t)) <= RTOL_LL * abs(after)
    y[40:60] *= 2.0  # exact: a factor of two
    assert mc.log_posterior(t, *args) == before
    # threads: every call sees the arrays it was given
    got = {}

    def worker(i):
        got[i] = [mc.log_posterior(tuple(th), *args) for th in live]

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert all(got[i] == got[0] for i in range(4)) and got[0][0] == before


# ------------------------------------------------------------------------------------------------
# f4: Lyman-limit-system profile (reference voigt_lls.py) behind DLAGP_FLAG_LLS_BREAK
# ------------------------------------------------------------------------------------------------
def test_lls_profile_matches_reference_golden_and_oracle():
    from oracle import ref_like as R

    g = load_golden("voigt_lls.npz")
    grid = g["grid"]
    n_u = grid.shape[0] - 6
    rng = np.random.default_rng(3)
    keep = np.ones(n_u, dtype=bool)
    keep[rng.choice(n_u, 20, replace=False)] = False
    n = int(keep.sum())
    from gpy_dla_detection_amd.sightline import DeviceSightline

    y, mu = 1 + 0.1 * rng.standard_normal(n), np.ones(n)
    M = 0.1 * rng.standard_normal((n, 20))
    omega2, v = 0.01 * (1 + rng.random(n)), 0.02 * (1 + rng.random(n))
    for broadening in (True, False):
        wl = grid if broadening else grid[3:-3]
        dev = DeviceSightline(y, mu, M, omega2, v, wl, keep, broadening=broadening, lls_break=True)
        want = g["absorption_b%d" % broadening]
        want = want if broadening else want[:, 3:-3]
        for i, (z, ln) in enumerate(zip(g["z"], g["log_nhi"])):
            np.testing.assert_allclose(dev.absorption([z], [10.0 ** ln]), want[i][keep], rtol=0, atol=RTOL_ABS)
        # log-likelihoods with the break, two absorbers: device vs oracle
        st = R.SpectrumState(y=y, v=v, this_mu=mu, this_M=M, this_omega2=omega2, padded_wavelengths=grid,
                             unmasked_wavelengths=grid[3:-3], keep=keep, broadening=broadening, lls_break=True)
        z2 = np.stack([g["z"][:6], g["z"][6:]], axis=1)
        n2 = 10.0 ** np.stack([g["log_nhi"][:6], g["log_nhi"][6:]], axis=1)
        got = dev.eval(z2, n2)
        from gpy_dla_detection_amd.sightline import get_option

        assert get_option("last_launch_path") == 2  # several Lyman-limit absorbers per sample: MFMA kernel too
        ref = np.array([R.sample_log_likelihood_k_dlas(st, z2[i], n2[i]) for i in range(6)])
        np.testing.assert_allclose(got, ref, rtol=RTOL_LL)
        z3 = np.stack([g["z"][:4], g["z"][4:8], g["z"][8:12]], axis=1)
        n3 = 10.0 ** np.stack([g["log_nhi"][:4], g["log_nhi"][4:8], g["log_nhi"][8:12]], axis=1)
        ref3 = np.array([R.sample_log_likelihood_k_dlas(st, z3[i], n3[i]) for i in range(4)])
        np.testing.assert_allclose(dev.eval(z3, n3), ref3, rtol=RTOL_LL)
        plain = DeviceSightline(y, mu, M, omega2, v, wl, keep, broadening=broadening).eval(z2, n2)
        assert np.max(np.abs(plain - got)) > 1.0  # the break matters on this grid
        # one absorber per sample on this log-uniform grid: the MFMA kernel with the continuum term
        z1, n1 = g["z"], 10.0 ** g["log_nhi"]
        got1 = dev.eval(z1, n1)
        ref1 = np.array([R.sample_log_likelihood_k_dlas(st, [z1[i]], [n1[i]]) for i in range(len(z1))])
        np.testing.assert_allclose(got1, ref1, rtol=RTOL_LL)
        from gpy_dla_detection_amd.sightline import set_option
        set_option("path", 1)
        try:
            generic1 = dev.eval(z1, n1)
        finally:
            set_option("path", 0)
        np.testing.assert_allclose(got1, generic1, rtol=1e-11)
        if broadening:  # and the device-resident sweep on top of it (levels > 1 multiply cached LLS profiles)
            from gpy_dla_detection_amd import parallel
            from gpy_dla_detection_amd.sightline import DeviceBatch

            S = 160
            zq = 3.05 + 1.2 * halton(S, 2)
            nq = 10.0 ** (17.0 + 5.5 * halton(S, 3))
            batch = DeviceBatch([dev])
            want = parallel.sweep_batch(batch.eval, zq[None, :], nq, 3, 0.01, seed=4)
            have = parallel.sweep_batch_device(batch, zq[None, :], nq, 3, 0.01, seed=4)
            np.testing.assert_array_equal(have[2], want[2])
            np.testing.assert_allclose(have[0], want[0], rtol=1e-10)
            np.testing.assert_allclose(have[1], want[1], rtol=1e-10, equal_nan=True)
    with pytest.raises(ValueError):
        from gpy_dla_detection_amd import _lib
        import ctypes

        h = ctypes.c_void_p()
        k8 = np.ascontiguousarray(keep, dtype=np.uint8)
        _lib.check(_lib.load().dlagp_create(ctypes.byref(h), n, n_u, 20, _lib.ptr(y), _lib.ptr(mu), _lib.ptr(M), 20, 1,
                                            _lib.ptr(omega2), _lib.ptr(v), _lib.ptr(grid),
                                            k8.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), 3, 8, 0))  # unknown flag bit


def test_lls_gp_class_runs_the_sweep_with_the_break():
    from gpy_dla_detection_amd.lls_gp import LLSGP

    sight = make_sightline(1, n_unmasked=400, n_masked=9)
    _, _, dla = build_models(sight, 64)
    lls = LLSGP.__new__(LLSGP)
    lls.__dict__.update(dla.__dict__)
    lls._device_sightline, lls._device_key = None, None
    assert lls.device_sightline().lls_break and not dla.device_sightline().lls_break
    z, _, nhi = make_qmc_samples(sight, 4)
    # this synthetic grid lies redwards of every absorber's limit: the break adds nothing there
    np.testing.assert_allclose(lls.device_sightline().eval(z, nhi), dla.device_sightline().eval(z, nhi), rtol=1e-11)


def test_core_patch_equals_in_loop_core_test(restore_options):
    """
    kd = 1 on a log-uniform grid runs the MFMA kernel that repairs the line cores outside its sample loop;
    option core_patch = 0 keeps the test inside the loop.  Same numbers, including samples whose line
    centres sit on the first / last entries, outside the grid, and NaN / infinite redshifts.
    """
    from oracle import ref_like as R

    set_option = restore_options
    sight = make_sightline(2)
    dev = device_sightline(sight)
    z, _, nhi = make_qmc_samples(sight, 4096)
    lam = sight.padded_wavelengths
    lines = np.array([1215.6701, 1025.7223, 972.5368])
    special = []
    for e in (0, 1, 2, 3, 4, 5, 6, lam.shape[0] - 1, lam.shape[0] - 2, lam.shape[0] - 4, 1500):
        for line in lines:                       # line centre exactly on / between chosen entries
            special += [lam[e] / line - 1.0, np.sqrt(lam[e] * lam[min(e + 1, lam.shape[0] - 1)]) / line - 1.0]
    special += [lam[0] / lines[0] - 1.0 - 0.002, lam[-1] / lines[0] - 1.0 + 0.002, 0.0, 9.0, -0.5, np.nan, np.inf]
    z = np.concatenate([z, special])
    nhi = np.concatenate([nhi, np.resize([1e20, 3e21, 1e23], len(special))])
    set_option("path", 2)
    got = dev.eval(z, nhi)
    set_option("core_patch", 0)
    want = dev.eval(z, nhi)
    finite = np.isfinite(want)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    # the repaired entries carry two roundings of 2^t (|t| up to ~1e3) instead of one: ~1e-12 in the total
    np.testing.assert_allclose(got[finite], want[finite], rtol=2e-11)
    assert abs(got[-2] - dev.null()) <= 1e-12 * abs(dev.null())   # NaN redshift: no absorber at all
    st = oracle_state(sight)
    pick = np.r_[0:8, len(z) - len(special):len(z) - 2]
    ref = np.array([R.sample_log_likelihood_k_dlas(st, [z[i]], [nhi[i]]) for i in pick])
    np.testing.assert_allclose(got[pick], ref, rtol=RTOL_LL)
    # several absorbers per sample: the cores are repaired pass by pass and absorber by absorber -- same numbers as
    # the in-loop test, with the special redshifts (edges of the grid, NaN, infinite) in every absorber position
    rng = np.random.default_rng(11)
    n_all = len(z)
    for kd in (2, 3, 4):
        idx = np.stack([np.arange(n_all)] + [rng.permutation(n_all) for _ in range(kd - 1)], axis=1)
        idx[-len(special):, 1] = np.arange(n_all - len(special), n_all)[::-1]  # special with special
        zz, nn = z[idx], nhi[idx]
        set_option("core_patch", 1)
        got = dev.eval(zz, nn)
        set_option("core_patch", 0)
        want = dev.eval(zz, nn)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        finite = np.isfinite(want)
        np.testing.assert_allclose(got[finite], want[finite], rtol=2e-11, err_msg="kd=%d" % kd)
        pick = np.r_[0:4, n_all - 6:n_all - 2]
        ref = np.array([R.sample_log_likelihood_k_dlas(st, zz[i], nn[i]) for i in pick])
        ok = np.isfinite(ref)
        np.testing.assert_allclose(got[pick][ok], ref[ok], rtol=RTOL_LL)
    # a grid that is not log-uniform silently takes the in-loop kernel
    set_option("core_patch", 1)
    warped = lam * (1 + 3e-4 * np.sin(np.arange(lam.shape[0]) / 90.0))
    from gpy_dla_detection_amd.sightline import DeviceSightline

    dev_w = DeviceSightline(sight.y, sight.this_mu, sight.this_M, sight.this_omega2, sight.v, warped, sight.keep)
    got_w = dev_w.eval(z[:64], nhi[:64])
    st.padded_wavelengths = warped
    ref_w = np.array([R.sample_log_likelihood_k_dlas(st, [z[i]], [nhi[i]]) for i in range(8)])
    np.testing.assert_allclose(got_w[:8], ref_w, rtol=RTOL_LL)


@pytest.mark.parametrize("S,max_dlas,nb", [(1, 1, 1), (5, 2, 1), (17, 3, 2), (64, 8, 1), (1030, 2, 3)])
def test_device_sweep_shapes(S, max_dlas, nb, restore_options):
    """odd sample counts (below / across the 16-sample groups and the 1024-thread scan), deep and shallow sweeps"""
    from gpy_dla_detection_amd import parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch

    sights = [make_sightline(10 + i, n_unmasked=230 + 17 * i, n_masked=3 + i) for i in range(nb)]
    batch = DeviceBatch([device_sightline(s) for s in sights])
    zs = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    nhi = make_qmc_samples(sights[0], S)[2]
    min_sep = 3000.0 * 1000.0 / 299792458.0
    want = parallel.sweep_batch(batch.eval, zs, nhi, max_dlas, min_sep, seed=11)
    got = parallel.sweep_batch_device(batch, zs, nhi, max_dlas, min_sep, seed=11)
    np.testing.assert_array_equal(got[2], want[2])
    assert np.array_equal(np.isnan(got[1]), np.isnan(want[1]))
    np.testing.assert_allclose(got[1], want[1], rtol=1e-11, equal_nan=True)
    np.testing.assert_allclose(got[0], want[0], rtol=1e-11, equal_nan=True)
    assert got[0].shape == (nb, max_dlas) and got[1].shape == (nb, S, max_dlas) and got[2].shape == (nb, max(max_dlas - 1, 0), S)


def test_class_api_device_levels_consume_the_rng_like_the_reference():
    """DLAGP.log_model_evidences: levels on the device (default) vs host levels -- same arrays, same RNG state after"""
    sight = make_sightline(1, n_unmasked=400, n_masked=9)
    _, _, dla = build_models(sight, 300)
    results = {}
    for host_levels in (False, True):
        dla.host_levels = host_levels
        np.random.seed(5)
        ev = dla.log_model_evidences(4)
        results[host_levels] = (ev, dla.sample_log_likelihoods.copy(), dla.base_sample_inds.copy(),
                                np.random.get_state()[1].copy(), np.random.get_state()[2])
    dev, host = results[False], results[True]
    np.testing.assert_array_equal(dev[2], host[2])
    np.testing.assert_allclose(dev[0], host[0], rtol=1e-12)
    np.testing.assert_allclose(dev[1], host[1], rtol=1e-12, equal_nan=True)
    assert np.array_equal(dev[3], host[3]) and dev[4] == host[4]      # legacy RNG left in the same state
    assert dev[1].shape == (300, 4) and dev[2].shape == (3, 300) and dev[2].dtype == np.int32
    # early stop (NaN evidence at the first level): nothing is drawn, as in the reference (dla_gp.py:200-206)
    good = dla.dla_samples._nhi_samples
    try:
        dla.dla_samples._nhi_samples = np.full(300, np.nan)
        for host_levels in (False, True):
            dla.host_levels = host_levels
            np.random.seed(9)
            before = np.random.get_state()[1].copy()
            ev = dla.log_model_evidences(3)
            assert np.isnan(ev).all() and np.array_equal(np.random.get_state()[1], before)
    finally:
        dla.dla_samples._nhi_samples = good


def test_randomised_sightlines_against_oracle(restore_options):
    """
    60 sightlines of random length, mask pattern (random single pixels, masked runs of 2..300 pixels), rank and
    redshift range, a few samples each with one or two absorbers; every sightline of rank <= 20 is required to run
    on the MFMA kernel (path = 2), whatever its mask.
    """
    from gpy_dla_detection_amd.sightline import get_option
    from oracle import ref_like as R

    set_option = restore_options
    rng = np.random.default_rng(2024)
    worst = 0.0
    for trial in range(60):
        n_u = int(rng.integers(40, 3200))
        n_masked = int(rng.integers(0, max(1, n_u // 12)))
        k = int(rng.choice([3, 20, 20, 20, 24]))
        sight = make_sightline(1000 + trial, n_unmasked=n_u, n_masked=n_masked, k=k)
        if trial % 7 == 3 and n_u > 200:  # a masked run inside the spectrum
            keep = sight.keep.copy()
            run = int(rng.integers(2, min(300, n_u // 2)))
            lo = int(rng.integers(10, max(11, n_u - run - 10)))
            sel = np.ones(int(keep.sum()), dtype=bool)
            idx = np.cumsum(keep) - 1
            gone = keep.copy()
            gone[:] = False
            gone[lo:lo + run] = keep[lo:lo + run]
            sel[idx[gone]] = False
            keep[lo:lo + run] = False
            sight.keep = keep
            sight.y, sight.v = sight.y[sel], sight.v[sel]
            sight.this_mu, sight.this_M, sight.this_omega2 = sight.this_mu[sel], sight.this_M[sel], sight.this_omega2[sel]
        dev = device_sightline(sight)
        S = 24
        z, _, nhi = make_qmc_samples(sight, S)
        z = z + 1e-3 * rng.standard_normal(S)
        set_option("path", 2)  # the MFMA kernel must serve it whatever the rank (raises otherwise)
        got = dev.eval(z, nhi)
        assert get_option("last_launch_path") == 2
        st = oracle_state(sight)
        pick = rng.choice(S, 6, replace=False)
        want = np.array([R.sample_log_likelihood_k_dlas(st, [z[i]], [nhi[i]]) for i in pick])
        rel = np.max(np.abs(got[pick] - want) / np.abs(want))
        if trial % 3 == 0:  # two absorbers per sample
            z2, n2 = np.stack([z, z[::-1]], axis=1), np.stack([nhi, nhi[::-1]], axis=1)
            got2 = dev.eval(z2, n2)
            want2 = np.array([R.sample_log_likelihood_k_dlas(st, z2[i], n2[i]) for i in pick[:3]])
            rel = max(rel, np.max(np.abs(got2[pick[:3]] - want2) / np.abs(want2)))
        set_option("path", 0)
        worst = max(worst, rel)
        assert rel <= RTOL_LL, "trial %d (n_u=%d, masked=%d, k=%d): %.2e" % (trial, n_u, n_masked, k, rel)
        assert abs(dev.null() - R.null_log_model_evidence(st)) <= RTOL_LL * abs(R.null_log_model_evidence(st))
    assert worst < RTOL_LL


def test_randomised_device_sweeps_against_the_oracle_driven_host_levels():
    """
    The whole device chain against the oracle: batches of 1..4 random sightlines of random flavour (3 / 5 / 1 lines, with and
    without broadening), random sample counts, depths 1..4, the levels chained on the device (accumulate, finalize,
    separation mask, evidence, resampling, profile cache) -- against `parallel.sweep_batch` around the ORACLE as evaluator
    (the reference's level logic in NumPy): same base indices, same NaN pattern, evidences and likelihoods to 1e-9.
    """
    from gpy_dla_detection_amd import parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch
    from oracle import ref_like as R

    rng = np.random.default_rng(31)
    flavours = [dict(), dict(), dict(broadening=False), dict(num_lines=5), dict(num_lines=1)]
    for trial in range(8):
        nb = int(rng.integers(1, 5))
        S = int(rng.integers(20, 90))
        max_dlas = int(rng.integers(1, 5))
        sights, states, devs = [], [], []
        for b in range(nb):
            n_u = int(rng.integers(80, 520))
            sight = make_sightline(2000 + 10 * trial + b, n_unmasked=n_u, n_masked=int(rng.integers(0, n_u // 10)))
            f = flavours[int(rng.integers(0, len(flavours)))]
            sights.append(sight)
            states.append(oracle_state(sight, num_lines=f.get("num_lines", 3), broadening=f.get("broadening", True)))
            devs.append(device_sightline(sight, **f))
        zs = np.stack([make_qmc_samples(s, S)[0] for s in sights])
        nhi = make_qmc_samples(sights[0], S)[2]
        min_sep = 3000.0e3 / 299792458.0

        def oracle_eval(z, n):
            return np.stack([np.array([R.sample_log_likelihood_k_dlas(st, z[b][i], n[b][i]) for i in range(z.shape[1])])
                             for b, st in enumerate(states)])

        want = parallel.sweep_batch(oracle_eval, zs, nhi, max_dlas, min_sep, seed=trial)
        batch = DeviceBatch(devs)
        got = parallel.sweep_batch_device(batch, zs, nhi, max_dlas, min_sep, seed=trial)
        batch.close()
        np.testing.assert_array_equal(got[2], want[2], err_msg="trial %d" % trial)
        assert np.array_equal(np.isnan(got[1]), np.isnan(want[1])), trial
        np.testing.assert_allclose(got[1], want[1], rtol=RTOL_LL, equal_nan=True, err_msg="trial %d" % trial)
        np.testing.assert_allclose(got[0], want[0], rtol=RTOL_LL, equal_nan=True, err_msg="trial %d" % trial)


def test_async_launches_survive_scratch_growth_and_destroy():
    """
    Stream rules of the *_device entry points (include/dlagp.h): a launch that needs larger scratch drains the stream
    of the launches before it, and destroy drains the last launch's stream before device memory goes back to the pool --
    results of back-to-back asynchronous launches with a growing sample count on a non-blocking stream stay right, and
    a context destroyed straight after an asynchronous launch does not corrupt a context created next.
    """
    import torch

    torch.cuda.init()
    dev = torch.device("cuda", 0)
    sight = make_sightline(3, n_unmasked=900, n_masked=17)
    z, _, nhi = make_qmc_samples(sight, 6000)
    want = device_sightline(sight).eval(z, nhi)
    stream = torch.cuda.Stream(device=dev)  # non-blocking with respect to the null stream
    d_z, d_n = torch.from_numpy(z).to(dev), torch.from_numpy(nhi).to(dev)
    torch.cuda.synchronize()
    ds = device_sightline(sight)
    outs = []
    with torch.cuda.stream(stream):
        for S in (64, 700, 6000, 300, 6000):  # the third launch regrows the partial-sum buffer while two are queued
            out = torch.full((S,), float("nan"), dtype=torch.float64, device=dev)
            ds.eval_device(S, 1, d_z.data_ptr(), d_n.data_ptr(), out.data_ptr(), stream.cuda_stream)
            outs.append((S, out))
    ds.close()  # straight after the last asynchronous launch
    other = device_sightline(make_sightline(4, n_unmasked=900, n_masked=17))  # draws the same pooled buffer sizes
    other_ll = other.eval(z[:500], nhi[:500])
    stream.synchronize()
    for S, out in outs:
        np.testing.assert_allclose(out.cpu().numpy(), want[:S], rtol=1e-13)
    again = device_sightline(make_sightline(4, n_unmasked=900, n_masked=17)).eval(z[:500], nhi[:500])
    np.testing.assert_array_equal(other_ll, again)
    # a member context destroyed before its batch: the batch's launches are drained first, later calls are refused
    from gpy_dla_detection_amd.sightline import DeviceBatch

    members = [device_sightline(sight), device_sightline(make_sightline(4, n_unmasked=900, n_masked=17))]
    batch = DeviceBatch(members)
    out2 = torch.full((2, 6000), float("nan"), dtype=torch.float64, device=dev)
    zz = torch.from_numpy(np.stack([z, z])).to(dev)
    nn = torch.from_numpy(np.stack([nhi, nhi])).to(dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        batch.eval_device(6000, 1, zz.data_ptr(), nn.data_ptr(), out2.data_ptr(), stream.cuda_stream)
    members[0].close()
    stream.synchronize()
    np.testing.assert_allclose(out2[0].cpu().numpy(), want, rtol=1e-13)
    with pytest.raises(ValueError):
        batch.eval_device(6000, 1, zz.data_ptr(), nn.data_ptr(), out2.data_ptr(), stream.cuda_stream)
    batch.close()
    members[1].close()


def test_async_device_sweeps_survive_growth_of_every_scratch_buffer():
    """
    The same stream rule for the multi-level device sweep (dlagp_batch_sweep_device): its per-level likelihoods, cdf
    scratch, alive flags, profile cache and running products all grow with S, and every growth drains the launches
    queued before it (one helper, `grow`, dlagp_hip.hip) -- back-to-back asynchronous sweeps with a growing sample
    count on a non-blocking stream give the results of synchronous sweeps on fresh batches.
    """
    import torch

    from gpy_dla_detection_amd import parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch

    dev = torch.device("cuda", 0)
    sights = [make_sightline(3, n_unmasked=700, n_masked=13), make_sightline(5, n_unmasked=640, n_masked=40)]
    sizes = (96, 1500, 400, 5000, 5000)
    max_dlas, min_sep = 3, 0.01
    nhi_all = make_qmc_samples(sights[0], max(sizes))[2]
    z_all = np.stack([make_qmc_samples(s, max(sizes))[0] for s in sights])
    want = {}
    for S in set(sizes):
        fresh = DeviceBatch([device_sightline(s) for s in sights])
        want[S] = fresh.sweep(z_all[:, :S], nhi_all[:S], max_dlas, min_sep, parallel.resampling_uniforms(2, max_dlas, S, seed=0))
        fresh.close()
    batch = DeviceBatch([device_sightline(s) for s in sights])
    stream = torch.cuda.Stream(device=dev)
    held = []
    for S in sizes:  # inputs first: the sweeps below are then enqueued back to back
        f64 = dict(dtype=torch.float64, device=dev)
        held.append(dict(
            S=S, z=torch.from_numpy(np.ascontiguousarray(z_all[:, :S])).to(dev), nhi=torch.from_numpy(nhi_all[:S].copy()).to(dev),
            u=torch.from_numpy(parallel.resampling_uniforms(2, max_dlas, S, seed=0)).to(dev),
            ev=torch.full((2, max_dlas), float("nan"), **f64), ll=torch.full((2, max_dlas, S), float("nan"), **f64),
            base=torch.zeros((2, max_dlas - 1, S), dtype=torch.int32, device=dev)))
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        for h in held:
            batch.sweep_device(h["S"], max_dlas, h["z"].data_ptr(), h["nhi"].data_ptr(), 0, min_sep, h["u"].data_ptr(),
                               h["ev"].data_ptr(), h["ll"].data_ptr(), h["base"].data_ptr(), stream.cuda_stream)
    stream.synchronize()
    for h in held:
        ev, ll, base = want[h["S"]]
        np.testing.assert_array_equal(h["base"].cpu().numpy(), base)
        np.testing.assert_allclose(h["ev"].cpu().numpy(), ev, rtol=1e-12)
        np.testing.assert_allclose(h["ll"].cpu().numpy().transpose(0, 2, 1), ll, rtol=1e-12, equal_nan=True)
    batch.close()


def test_both_finalisations_give_the_same_numbers(restore_options):
    """
    finalize_rank20 factors with 16 lanes per sample on launches of up to `finalize_row16_blocks` blocks and with one
    lane per sample beyond: every entry of the factor sees the same operations in the same order, so the two forms
    agree bit for bit (full BASELINE size, one and two absorbers, the non-PD flag included).
    """
    from gpy_dla_detection_amd.sightline import DeviceSightline, get_option, set_option

    sight = make_sightline(2)
    dev = device_sightline(sight)
    z, _, nhi = make_qmc_samples(sight, 10000)
    z2, n2 = np.stack([z, z[::-1]], axis=1)[:3000], np.stack([nhi, nhi[::-1]], axis=1)[:3000]
    default = get_option("finalize_row16_blocks")
    assert default >= 625
    try:
        wide, wide2 = dev.eval(z, nhi), dev.eval(z2, n2)
        set_option("finalize_row16_blocks", 0)
        serial, serial2 = dev.eval(z, nhi), dev.eval(z2, n2)
    finally:
        set_option("finalize_row16_blocks", default)
    np.testing.assert_array_equal(wide, serial)
    np.testing.assert_array_equal(wide2, serial2)
    st = oracle_state(sight)
    from oracle import ref_like as R

    for i in (0, 17, 4999, 9999):
        want = R.sample_log_likelihood_k_dlas(st, [z[i]], [nhi[i]])
        assert abs(wide[i] - want) <= RTOL_LL * abs(want)
    set_option("negative_noise_mfma", 1)
    neg = DeviceSightline(sight.y, sight.this_mu, 10.0 * sight.this_M, sight.this_omega2, -np.abs(sight.v),
                          sight.padded_wavelengths, sight.keep)
    set_option("negative_noise_mfma", 0)
    for blocks in (default, 0):
        set_option("finalize_row16_blocks", blocks)
        try:
            with pytest.raises(np.linalg.LinAlgError):
                neg.eval(z[:64], nhi[:64])
        finally:
            set_option("finalize_row16_blocks", default)


def test_accumulate_waves_finish_their_own_blocks_bit_for_bit(restore_options):
    """
    Launch plans with ONE piece per block (batches: more blocks than wave slots; any launch with pixel_splits = 1): the
    accumulate wave factors its own block (kernels_mfma.h FUSED) and no partial sums cross HBM.  The same operations in
    the same order as finalize_rank20's 16-lane form (one shared routine), and a single piece has nothing to add up: the
    results equal the two-kernel form's BIT FOR BIT -- one absorber, several absorbers per sample, every level of a
    cached and of a recomputed sweep (base indices and evidences with them), S not a multiple of 16, ragged sightlines,
    a by-sample slice -- and the non-PD flag is raised by the wave itself.
    """
    from gpy_dla_detection_amd import parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch, DeviceSightline, get_option, set_option

    sights = [make_sightline(60, n_unmasked=685, n_masked=14), make_sightline(61, n_unmasked=300, n_masked=6),
              make_sightline(62, n_unmasked=1250, n_masked=25), make_sightline(63, n_unmasked=58, n_masked=1)]
    devs = [device_sightline(s) for s in sights]
    batch = DeviceBatch(devs)
    S, max_dlas = 1003, 4
    zs = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    nhi = make_qmc_samples(sights[0], S)[2]
    uniforms = parallel.resampling_uniforms(len(sights), max_dlas, S, seed=9)
    z3 = np.stack([zs, zs[:, ::-1], np.roll(zs, 7, axis=1)], axis=2)
    n3 = np.stack([np.broadcast_to(nhi, zs.shape), np.broadcast_to(nhi[::-1], zs.shape), np.broadcast_to(np.roll(nhi, 7), zs.shape)], axis=2)
    set_option("pixel_splits", 1)  # one piece per block whatever the launch's size

    def everything():
        out = {"one": batch.eval(zs, np.broadcast_to(nhi, zs.shape)), "three": batch.eval(z3, n3)}
        fused = [get_option("last_launch_fused")]
        for cache in (1, 0):
            set_option("profile_cache", cache)
            ev, ll, base = batch.sweep(zs, nhi, max_dlas, 0.002, uniforms)
            fused.append(get_option("last_launch_fused"))
            out["sweep%d" % cache] = (ev, np.array(ll), base)
        set_option("profile_cache", 1)
        out["single"] = devs[0].eval(zs[0], nhi)
        return out, fused

    set_option("fuse_finalize", 1)
    got, fused = everything()
    assert fused == [1, 1, 1]
    set_option("fuse_finalize", 0)
    want, unfused = everything()
    assert unfused == [0, 0, 0]
    set_option("fuse_finalize", 1)
    for key in want:
        if isinstance(want[key], tuple):
            for a, b in zip(got[key], want[key]):
                np.testing.assert_array_equal(a, b, err_msg=key)
        else:
            np.testing.assert_array_equal(got[key], want[key], err_msg=key)
    # the fused numbers against the oracle, directly
    from oracle import ref_like as R

    for b in (0, 3):
        st = oracle_state(sights[b])
        for i in (0, 500, S - 1):
            want_ll = R.sample_log_likelihood_k_dlas(st, z3[b, i], n3[b, i])
            assert abs(got["three"][b, i] - want_ll) <= RTOL_LL * abs(want_ll)
    # automatic plans: a launch with more blocks than wave slots is single-piece and fused, a small one is not
    set_option("pixel_splits", 0)
    big = 8192  # 4 x 512 blocks = the chip's 2048 wave slots: one piece per block is the planner's own choice
    zb = np.stack([make_qmc_samples(s, big)[0] for s in sights])
    nb_ = make_qmc_samples(sights[0], big)[2]
    first = batch.eval(zb, np.broadcast_to(nb_, zb.shape))
    assert get_option("last_launch_parts") == 1 and get_option("last_launch_fused") == 1
    devs[0].eval(zs[0], nhi)
    assert get_option("last_launch_parts") > 1 and get_option("last_launch_fused") == 0
    set_option("fuse_finalize", 0)
    np.testing.assert_array_equal(first, batch.eval(zb, np.broadcast_to(nb_, zb.shape)))
    set_option("fuse_finalize", 1)
    # a matrix that is not positive definite: the wave raises the flag (LinAlgError, numpy's at null_gp.py:343)
    set_option("negative_noise_mfma", 1)
    neg = DeviceSightline(sights[0].y, sights[0].this_mu, 10.0 * sights[0].this_M, sights[0].this_omega2,
                          -np.abs(sights[0].v), sights[0].padded_wavelengths, sights[0].keep)
    set_option("negative_noise_mfma", 0)
    set_option("pixel_splits", 1)
    with pytest.raises(np.linalg.LinAlgError):
        neg.eval(zs[0, :64], nhi[:64])
    assert get_option("last_launch_fused") == 1
    # ... and the next launch on a healthy sightline starts from a clear flag
    np.testing.assert_array_equal(devs[0].eval(zs[0], nhi), got["single"])
    neg.close()
    batch.close()


def test_sub_dla_and_dla_models_share_one_optical_depth_in_a_pair_launch(restore_options):
    """
    The sub-DLA model and level 1 of the DLA model are one absorber per sample at the SAME redshift samples and differ
    only in the column density (MATLAB: one `sample_z_dlas(i)` for both, multi_dlas/process_qsos_multiple_dlas_meanflux.m:
    335-373; Python: dla_samples.py:94-104 = subdla_samples.py:115-125).  When the samples are equal by content and the
    batch's shape allows, ONE launch evaluates both -- a wave's 16 MFMA rows are 8 QMC points x 2 column densities, the
    Voigt optical depth is formed once per point (kernels_mfma.h PAIR).  Every sample's sums are formed exactly as in a
    launch of its own, so with equal cuts (one piece per block here) the pair launch equals the two separate sweeps BIT FOR
    BIT: evidences, per-sample likelihoods, base indices, NaN pattern -- the device call and the host call, cached and
    recomputed deeper levels, max_dlas 1 and 4, S not a multiple of 8, ragged sightlines; different redshift sets fall
    back to two launches.  And both agree with the oracle.
    """
    import torch

    from gpy_dla_detection_amd import parallel
    from gpy_dla_detection_amd.sightline import DeviceBatch, get_option, set_option

    sights = [make_sightline(70, n_unmasked=685, n_masked=14), make_sightline(71, n_unmasked=300, n_masked=6),
              make_sightline(72, n_unmasked=1250, n_masked=25), make_sightline(73, n_unmasked=58, n_masked=1)]
    devs = [device_sightline(s) for s in sights]
    batch = DeviceBatch(devs)
    nb = len(sights)
    set_option("pixel_splits", 1)  # one piece per block: every launch of this test is finished by its own waves
    for S, max_dlas in ((1003, 4), (64, 1), (1000, 2)):
        zs = np.stack([make_qmc_samples(s, S)[0] for s in sights])
        nhi = make_qmc_samples(sights[0], S)[2]
        nhi_sub = make_qmc_samples(sights[0], S, sub_dla=True)[2]
        uniforms = parallel.resampling_uniforms(nb, max_dlas, S, seed=11) if max_dlas > 1 else None
        for cache in (1, 0):
            set_option("profile_cache", cache)
            set_option("pair_models", 1)
            got = batch.model_evidences(zs, nhi_sub, zs, nhi, max_dlas, 0.002, uniforms)
            assert get_option("last_sweep_paired") == 1
            set_option("pair_models", 0)
            want = batch.model_evidences(zs, nhi_sub, zs, nhi, max_dlas, 0.002, uniforms)
            assert get_option("last_sweep_paired") == 0
            np.testing.assert_array_equal(got[0], want[0])
            for model in (1, 2):
                for a, b in zip(got[model], want[model]):
                    np.testing.assert_array_equal(a, b, err_msg="S=%d max_dlas=%d cache=%d model=%d" % (S, max_dlas, cache, model))
        set_option("profile_cache", 1)
        # three equal pieces per block: the pair launch leaves partial sums and finalize_rank20 sorts the rows to the two
        # models -- the same cuts for 2 S rows as for S, so still bit for bit
        set_option("pixel_splits", 3)
        set_option("pair_models", 1)
        cut3 = batch.model_evidences(zs, nhi_sub, zs, nhi, max_dlas, 0.002, uniforms)
        assert get_option("last_sweep_paired") == 1 and get_option("last_launch_parts") in (1, 3)
        set_option("pair_models", 0)
        cut3_want = batch.model_evidences(zs, nhi_sub, zs, nhi, max_dlas, 0.002, uniforms)
        for model in (1, 2):
            for a, b in zip(cut3[model], cut3_want[model]):
                np.testing.assert_array_equal(a, b, err_msg="three pieces, S=%d model=%d" % (S, model))
        set_option("pixel_splits", 1)
        set_option("pair_models", 1)
        # the two-phase form resumes the paired sweep
        batch.model_evidences_start(zs, nhi_sub, zs, nhi, max_dlas, 0.002)
        two_phase = batch.model_evidences_finish(uniforms)
        for model in (1, 2):
            for a, b in zip(two_phase[model], got[model]):
                np.testing.assert_array_equal(a, b)
        # other redshifts for the sub-DLA model: two launches, the same answer as two sweeps
        other = batch.model_evidences(zs[:, ::-1].copy(), nhi_sub, zs, nhi, max_dlas, 0.002, uniforms)
        assert get_option("last_sweep_paired") == 0
        np.testing.assert_array_equal(other[2][0], got[2][0])
    # against the oracle: both models of the pair launch
    from oracle import ref_like as R

    S = 1003
    zs = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    nhi = make_qmc_samples(sights[0], S)[2]
    nhi_sub = make_qmc_samples(sights[0], S, sub_dla=True)[2]
    got = batch.model_evidences(zs, nhi_sub, zs, nhi, 1, 0.002, None)
    assert get_option("last_sweep_paired") == 1
    for b in (0, 3):
        st = oracle_state(sights[b])
        for i in (0, 7, 8, 500, S - 1):
            for model, table in ((1, nhi_sub), (2, nhi)):
                want_ll = R.sample_log_likelihood_k_dlas(st, [zs[b, i]], [table[i]]) - np.log(S)
                assert abs(got[model][1][b, i, 0] - want_ll) <= RTOL_LL * abs(want_ll)
    # the device-pointer call, automatic plans (4 x 8192 samples fill the chip with single-piece blocks)
    set_option("pixel_splits", 0)
    S, max_dlas = 8192, 2
    dev = torch.device("cuda", 0)
    zs = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    z = torch.from_numpy(zs).to(dev)
    t_nhi = torch.from_numpy(make_qmc_samples(sights[0], S)[2]).to(dev)
    t_sub = torch.from_numpy(make_qmc_samples(sights[0], S, sub_dla=True)[2]).to(dev)
    u = torch.from_numpy(parallel.resampling_uniforms(nb, max_dlas, S, seed=3)).to(dev)
    f64 = dict(dtype=torch.float64, device=dev)

    def device_run(pair):
        set_option("pair_models", pair)
        ev_sub, ll_sub = torch.zeros(nb, **f64), torch.zeros((nb, 1, S), **f64)
        ev, ll = torch.zeros((nb, max_dlas), **f64), torch.zeros((nb, max_dlas, S), **f64)
        base = torch.zeros((nb, max_dlas - 1, S), dtype=torch.int32, device=dev)
        batch.two_model_sweep_device(S, max_dlas, z.data_ptr(), t_sub.data_ptr(), 0, t_nhi.data_ptr(), 0, 0.002, u.data_ptr(),
                                     ev_sub.data_ptr(), ll_sub.data_ptr(), ev.data_ptr(), ll.data_ptr(), base.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return [t.cpu().numpy() for t in (ev_sub, ll_sub, ev, ll, base)], get_option("last_sweep_paired")

    paired, was_paired = device_run(1)
    separate, was_not = device_run(0)
    assert was_paired == 1 and was_not == 0
    for a, b in zip(paired[:4], separate[:4]):  # automatic plans may cut the separate launches elsewhere: rounding
        np.testing.assert_allclose(a, b, rtol=1e-12, equal_nan=True)
    assert np.mean(paired[4] == separate[4]) > 0.999
    batch.close()


def test_long_sightline_and_concurrent_context_creation():
    """
    Context creation stages its uploads in a page-locked arena of the calling thread (one synchronisation per context):
    a sightline three times the BASELINE length makes the arena grow mid-sequence, and contexts built by several
    threads at once (the catalogue driver's workers) stay independent.  Both against the oracle.
    """
    import threading

    from oracle import ref_like as R

    big = make_sightline(6, n_unmasked=9100, n_masked=140)
    dev = device_sightline(big)
    z, _, nhi = make_qmc_samples(big, 48)
    got = dev.eval(z, nhi)
    st = oracle_state(big)
    for i in (0, 11, 47):
        want = R.sample_log_likelihood_k_dlas(st, [z[i]], [nhi[i]])
        assert abs(got[i] - want) <= RTOL_LL * abs(want)
    assert abs(dev.null() - R.null_log_model_evidence(st)) <= RTOL_LL * abs(R.null_log_model_evidence(st))
    dev.close()

    sights = [make_sightline(10 + i, n_unmasked=500 + 37 * i, n_masked=5 + i) for i in range(12)]
    results, errors = [None] * len(sights), []

    def work(lo, hi):
        try:
            for i in range(lo, hi):
                for _ in range(3):  # create / evaluate / destroy: the pool hands buffers from thread to thread
                    d = device_sightline(sights[i])
                    zz, _, nn = make_qmc_samples(sights[i], 32)
                    results[i] = (zz, nn, d.eval(zz, nn), d.null())
                    d.close()
        except Exception as exc:  # surfaced below: an assertion inside a thread would otherwise vanish
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(lo, lo + 3)) for lo in range(0, len(sights), 3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for sight, (zz, nn, ll, null) in zip(sights, results):
        st = oracle_state(sight)
        want = np.array([R.sample_log_likelihood_k_dlas(st, [zz[i]], [nn[i]]) for i in (0, 31)])
        np.testing.assert_allclose(ll[[0, 31]], want, rtol=RTOL_LL)
        assert abs(null - R.null_log_model_evidence(st)) <= RTOL_LL * abs(null)


def test_host_pointer_calls_after_device_launches_on_another_stream():
    """
    The host-pointer entry points run on a stream of the calling thread.  Called on a batch whose `*_device` launches are
    still in flight on the caller's stream (bench.py config 4: two asynchronous sweeps, then `batch.nulls()`), they first
    drain that stream -- the launches share the batch's scratch -- and the next device launch drains theirs.
    """
    import torch

    from gpy_dla_detection_amd.sightline import DeviceBatch

    dev = torch.device("cuda", 0)
    sights = [make_sightline(20 + i, n_unmasked=900 + 50 * i, n_masked=10) for i in range(3)]
    batch = DeviceBatch([device_sightline(s) for s in sights])
    S = 4000
    zs = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    nhi = make_qmc_samples(sights[0], S)[2]
    want = batch.eval(zs, np.tile(nhi, (3, 1)))
    want_nulls = batch.nulls()
    stream = torch.cuda.Stream(device=dev)
    d_z = torch.from_numpy(zs).to(dev)
    d_n = torch.from_numpy(np.tile(nhi, (3, 1))).to(dev)
    outs = [torch.full((3, S), float("nan"), dtype=torch.float64, device=dev) for _ in range(3)]
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        batch.eval_device(S, 1, d_z.data_ptr(), d_n.data_ptr(), outs[0].data_ptr(), stream.cuda_stream)
        batch.eval_device(S, 1, d_z.data_ptr(), d_n.data_ptr(), outs[1].data_ptr(), stream.cuda_stream)
        nulls = batch.nulls()                      # host-pointer call, another stream: drains `stream` first
        small = batch.eval(zs[:, :64], np.tile(nhi[:64], (3, 1)))
        batch.eval_device(S, 1, d_z.data_ptr(), d_n.data_ptr(), outs[2].data_ptr(), stream.cuda_stream)
    stream.synchronize()
    np.testing.assert_array_equal(nulls, want_nulls)
    np.testing.assert_allclose(small, want[:, :64], rtol=1e-13)
    for out in outs:
        np.testing.assert_array_equal(out.cpu().numpy(), want)
    batch.close()


def test_device_entry_points_are_graph_capturable():
    """
    dlagp_eval_device + dlagp_level_evidence_device only enqueue work on the caller's stream once the workspace
    exists (no allocation, no synchronisation): a HIP graph captured around them replays to the same numbers.
    """
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch has no GPU")
    from gpy_dla_detection_amd import sightline as sl

    device = torch.device("cuda", 0)
    sight = make_sightline(1, n_unmasked=900, n_masked=11)
    dev = device_sightline(sight)
    S = 2000
    z, _, nhi = make_qmc_samples(sight, S)
    d_z, d_n = torch.from_numpy(z).to(device), torch.from_numpy(nhi).to(device)
    out = torch.empty(S, dtype=torch.float64, device=device)
    ev = torch.zeros(1, dtype=torch.float64, device=device)

    def step(stream):
        dev.eval_device(S, 1, d_z.data_ptr(), d_n.data_ptr(), out.data_ptr(), stream)
        sl.level_evidence_device(out.data_ptr(), 1, S, 0, ev.data_ptr(), device=0, stream=stream)

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        step(side.cuda_stream)  # first call sizes the workspace
        torch.cuda.synchronize()
        want, want_ev = out.clone(), ev.clone()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            step(torch.cuda.current_stream().cuda_stream)
    out.zero_()
    ev.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want) and torch.equal(ev, want_ev)
    np.testing.assert_allclose(out.cpu().numpy() + np.log(S), dev.eval(z, nhi), rtol=1e-13)


def test_level_evidence_from_the_finalisation_kernels_tail(restore_options):
    """
    dlagp_eval_evidence_device = dlagp_eval_device + dlagp_level_evidence_device (reference dla_gp.py:132-190 for one level) in
    one call.  Where the launch ends in finalize_rank20 the evidence comes from that kernel's tail -- the last workgroup to
    arrive combines the blocks' (max, sum, count) -- and no second launch follows: the log-likelihoods are the two calls'
    bit for bit (minus log S, NaN where masked), the evidence agrees with theirs to rounding and with NumPy's log-mean-exp,
    the same bits run after run; sample counts that do not fill the last block, a masked level, an all-NaN level, a
    not-positive-definite sample; option fold_evidence = 0 is the two launches exactly.  A batch of three sightlines draws one
    ticket counter per member.
    """
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch has no GPU")
    from gpy_dla_detection_amd import sightline as sl
    from gpy_dla_detection_amd.sightline import DeviceBatch, get_option, set_option

    device = torch.device("cuda", 0)
    rng = np.random.default_rng(5)

    def both_ways(target, nb, S, kd, z, nhi, level, mask):
        d_z, d_n = torch.from_numpy(z).to(device), torch.from_numpy(nhi).to(device)
        d_mask = torch.from_numpy(mask).to(device) if mask is not None else None
        mask_ptr = d_mask.data_ptr() if d_mask is not None else 0
        want = torch.empty((nb, S), dtype=torch.float64, device=device)
        want_ev = torch.zeros(nb, dtype=torch.float64, device=device)
        target.eval_device(S, kd, d_z.data_ptr(), d_n.data_ptr(), want.data_ptr())
        sl.level_evidence_device(want.data_ptr(), nb, S, level, want_ev.data_ptr(), d_too_close=mask_ptr)
        got = torch.full((nb, S), 7.0, dtype=torch.float64, device=device)
        got_ev = torch.full((nb,), 7.0, dtype=torch.float64, device=device)
        target.eval_evidence_device(S, kd, d_z.data_ptr(), d_n.data_ptr(), got.data_ptr(), level, got_ev.data_ptr(), d_too_close=mask_ptr)
        torch.cuda.synchronize()
        return want.cpu().numpy(), want_ev.cpu().numpy(), got.cpu().numpy(), got_ev.cpu().numpy()

    def numpy_evidence(ll, level, S):
        out = np.full(ll.shape[0], np.nan)
        for b, row in enumerate(ll):
            if np.isfinite(row).any():
                top = np.nanmax(row)
                out[b] = top + np.log(np.nanmean(np.exp(row - top))) - level * np.log(S)
        return out

    sight = make_sightline(3, n_unmasked=1400, n_masked=23)
    dev = device_sightline(sight)
    for S, kd, level, masked in ((10000, 1, 0, False), (1003, 2, 1, True), (17, 1, 0, False), (4096, 3, 2, True)):
        z, _, nhi = make_qmc_samples(sight, S * kd)
        z, nhi = z.reshape(S, kd).copy(), nhi.reshape(S, kd).copy()
        mask = (rng.random((1, S)) < 0.2).astype(np.uint8) if masked else None
        want, want_ev, got, got_ev = both_ways(dev, 1, S, kd, z, nhi, level, mask)
        assert get_option("last_evidence_folded") == 1, (S, kd)
        np.testing.assert_array_equal(got, want)  # NaN == NaN here
        if masked:
            assert np.isnan(got[0][mask[0] != 0]).all() and np.isfinite(got[0][mask[0] == 0]).all()
        np.testing.assert_allclose(got_ev, want_ev, rtol=1e-13)
        np.testing.assert_allclose(got_ev, numpy_evidence(got, level, S), rtol=1e-13)
        # the same bits run after run, whichever workgroup arrives last
        d_z, d_n = torch.from_numpy(z).to(device), torch.from_numpy(nhi).to(device)
        d_mask = torch.from_numpy(mask).to(device) if masked else None
        out = torch.empty((1, S), dtype=torch.float64, device=device)
        evs = torch.zeros(24, dtype=torch.float64, device=device)
        for i in range(24):
            dev.eval_evidence_device(S, kd, d_z.data_ptr(), d_n.data_ptr(), out.data_ptr(), level, evs.data_ptr() + 8 * i,
                                     d_too_close=d_mask.data_ptr() if masked else 0)
        torch.cuda.synchronize()
        assert np.unique(evs.cpu().numpy().view(np.int64)).size == 1
        np.testing.assert_array_equal(evs.cpu().numpy()[:1], got_ev)
        # launches that alternate between two masks: every evidence is its own launch's, never a block's sums left over from
        # the launch before (the blocks' triples cross compute units through write-through stores and a ticket, no fence)
        other = (rng.random((1, S)) < 0.5).astype(np.uint8)
        d_other = torch.from_numpy(other).to(device)
        _, _, _, other_ev = both_ways(dev, 1, S, kd, z, nhi, level, other)
        evs = torch.zeros(40, dtype=torch.float64, device=device)
        for i in range(40):
            use = d_other.data_ptr() if i % 2 else (d_mask.data_ptr() if masked else 0)
            dev.eval_evidence_device(S, kd, d_z.data_ptr(), d_n.data_ptr(), out.data_ptr(), level, evs.data_ptr() + 8 * i, d_too_close=use)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(evs.cpu().numpy()[0::2], np.repeat(got_ev, 20))
        np.testing.assert_array_equal(evs.cpu().numpy()[1::2], np.repeat(other_ev, 20))
        assert other_ev[0] != got_ev[0]
        # the two launches behind one call
        set_option("fold_evidence", 0)
        _, _, apart, apart_ev = both_ways(dev, 1, S, kd, z, nhi, level, mask)
        assert get_option("last_evidence_folded") == 0
        set_option("fold_evidence", 1)
        np.testing.assert_array_equal(apart, want)
        np.testing.assert_array_equal(apart_ev, want_ev)

    # the kernel's other forms carry the same tail: one lane per sample instead of sixteen (identical bits), and the generic
    # kernel's partial sums (another kernel's rounding: agreement with its own two launches)
    S = 1003
    z, _, nhi = make_qmc_samples(sight, S)
    mask = (rng.random((1, S)) < 0.2).astype(np.uint8)
    _, _, sixteen, sixteen_ev = both_ways(dev, 1, S, 1, z, nhi, 0, mask)
    default = get_option("finalize_row16_blocks")
    set_option("finalize_row16_blocks", 0)
    want, want_ev, one_lane, one_lane_ev = both_ways(dev, 1, S, 1, z, nhi, 0, mask)
    set_option("finalize_row16_blocks", default)
    assert get_option("last_evidence_folded") == 1
    np.testing.assert_array_equal(one_lane, sixteen)
    np.testing.assert_array_equal(one_lane_ev, sixteen_ev)
    set_option("path", 1)
    want, want_ev, generic, generic_ev = both_ways(dev, 1, 200, 1, z[:200], nhi[:200], 0, mask[:, :200].copy())
    assert get_option("last_launch_path") == 1 and get_option("last_evidence_folded") == 1
    set_option("path", 0)
    np.testing.assert_array_equal(generic, want)
    np.testing.assert_allclose(generic_ev, want_ev, rtol=1e-13)
    np.testing.assert_allclose(generic + np.log(200), sixteen[:, :200] + np.log(S), rtol=1e-9)  # (each minus its own log S)

    # every sample masked: an all-NaN level has a NaN evidence (np.nanmax / np.nanmean of nothing)
    S = 500
    z, _, nhi = make_qmc_samples(sight, S)
    want, want_ev, got, got_ev = both_ways(dev, 1, S, 1, z, nhi, 0, np.ones((1, S), dtype=np.uint8))
    assert np.isnan(got).all() and np.isnan(got_ev).all() and np.isnan(want_ev).all()

    # three sightlines of different lengths: one ticket counter each, evidences at the members' positions
    sights = [make_sightline(40 + i, n_unmasked=n_u, n_masked=m) for i, (n_u, m) in enumerate([(700, 3), (1250, 40), (300, 0)])]
    devs = [device_sightline(s) for s in sights]
    batch = DeviceBatch(devs)
    S = 1500
    zs = np.stack([make_qmc_samples(s, S)[0] for s in sights])
    nhi = np.stack([make_qmc_samples(s, S)[2] for s in sights])
    mask = (rng.random((3, S)) < 0.1).astype(np.uint8)
    want, want_ev, got, got_ev = both_ways(batch, 3, S, 1, zs, nhi, 0, mask)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_allclose(got_ev, want_ev, rtol=1e-13)
    np.testing.assert_allclose(got_ev, numpy_evidence(got, 0, S), rtol=1e-13)
    for b, d in enumerate(devs):  # ... and each member on its own (its launch is cut at other pixels: agreement to rounding)
        one = both_ways(d, 1, S, 1, zs[b], nhi[b], 0, mask[b:b + 1])
        np.testing.assert_allclose(one[2][0], got[b], rtol=1e-11)
        np.testing.assert_allclose(one[3][0], got_ev[b], rtol=1e-11)
    batch.close()
    for d in devs + [dev]:
        d.close()
