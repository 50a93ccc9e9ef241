"""
Pins the oracle (oracle/ref_like.py) against golden vectors produced by the reference
itself (tools/gen_golden.py) and against the reference's own known answers
(reference tests/test_model.py:52-72, tests/test_voigt.py:8-57).  CPU only.
"""
import numpy as np
import pytest
from scipy.stats import multivariate_normal

from gpy_dla_detection_amd import lyman_series as ls
from gpy_dla_detection_amd.synthetic import make_sightline
from oracle import ref_like as R
from tests.conftest import load_golden, oracle_state
from tests.golden_inputs import MVNPDF_SHAPES, checksum, mvnpdf_inputs

# oracle and reference call the same BLAS/LAPACK/wofz entry points in the same order,
# so agreement is expected at rounding level; 1e-12 relative leaves room for BLAS threading.
RTOL = 1e-12


def test_voigt_profile_matches_reference_grid():
    g = load_golden("voigt_profile.npz")
    for line in range(ls.MAX_LINES):
        mine = R.voigt_profile(g["velocity"], ls.SIGMA_CGS, ls.LORENTZ_WIDTHS[line])
        np.testing.assert_allclose(mine, g["profile"][line], rtol=1e-15, atol=0)


def test_voigt_absorption_matches_reference():
    g = load_golden("voigt_absorption.npz")
    n_cases = g["case_z"].shape[0]
    assert n_cases > 100
    for i in range(n_cases):
        grid = g["grid_" + str(g["case_grid"][i])]
        mine = R.voigt_absorption(
            grid,
            10.0 ** g["case_log_nhi"][i],
            g["case_z"][i],
            num_lines=int(g["case_num_lines"][i]),
            broadening=bool(g["case_broadening"][i]),
        )
        np.testing.assert_allclose(mine, g["out_%d" % i], rtol=1e-14, atol=1e-300)


def test_instrumental_broadening_is_valid_convolution():
    # reference tests/test_voigt.py:8-57: explicit 7-tap loop == np.convolve "valid"
    for z_qso, z_dla, log_nhi, lines, npts in ((3.15, 3.1, 20.3, 3, 1000), (5.0, 4.5, 21.0, 5, 50)):
        lam = np.linspace(911, 1216, npts) * (1 + z_qso)
        raw = R.voigt_absorption(lam, 10.0 ** log_nhi, z_dla, num_lines=lines, broadening=False)
        explicit = np.zeros(npts - 6)
        for i in range(npts - 6):
            for t in range(7):
                explicit[i] += raw[i + t] * ls.INSTRUMENT_PROFILE[t]
        smooth = R.voigt_absorption(lam, 10.0 ** log_nhi, z_dla, num_lines=lines, broadening=True)
        assert np.all(np.abs(explicit - smooth) < 1e-4)


def test_log_mvnpdf_known_answers():
    # reference tests/test_model.py:52-72, tolerance 1e-4 there
    mu = np.array([1, 2])
    M = np.array([[2, 3, 1], [1, 2, 4]])
    rv = multivariate_normal(mu, np.matmul(M, M.T) + np.eye(2) * 2)
    g = load_golden("mvnpdf.npz")
    for y, want in zip(([1, 2], [2, 3], [100, 100]), g["kat_out"]):
        got = R.log_mvnpdf_low_rank(np.array(y), mu, M, np.ones(2) * 2)
        assert abs(got - rv.logpdf(y)) < 1e-4
        assert abs(got - want) <= RTOL * abs(want)


@pytest.mark.parametrize("idx", range(len(MVNPDF_SHAPES)))
def test_log_mvnpdf_random_shapes(idx):
    g = load_golden("mvnpdf.npz")
    n, k = MVNPDF_SHAPES[idx]
    y, mu, M, d = mvnpdf_inputs(idx, n, k)
    assert checksum(y, mu, M, d) == int(g["rand_input_crc"][idx]), "regenerated inputs drifted"
    for col, lapack in ((0, True), (1, False)):
        got = R.log_mvnpdf_low_rank(y, mu, M, d, scipy_lapack=lapack)
        want = g["rand_out"][idx, col]
        assert abs(got - want) <= RTOL * abs(want)


def test_sample_log_likelihood_k_dlas():
    g = load_golden("sample_ll.npz")
    sight = make_sightline(int(g["sight_index"]))
    assert checksum(sight.y, sight.this_M, sight.padded_wavelengths, sight.keep) == int(g["sight_crc"])
    st = oracle_state(sight)
    assert abs(R.null_log_model_evidence(st) - g["null_evidence"]) <= RTOL * abs(g["null_evidence"])
    for kd in (1, 2, 3, 4):
        z, nhi, want = g["z_kd%d" % kd], g["nhi_kd%d" % kd], g["out_kd%d" % kd]
        got = np.array([R.sample_log_likelihood_k_dlas(st, z[i], nhi[i]) for i in range(0, z.shape[0], 4)])
        np.testing.assert_allclose(got, want[::4], rtol=RTOL)
    st_nb = oracle_state(sight, broadening=False)
    got = np.array([R.sample_log_likelihood_k_dlas(st_nb, g["z_kd1"][i], g["nhi_kd1"][i]) for i in range(0, 64, 8)])
    np.testing.assert_allclose(got, g["out_nobroad_kd1"][::8], rtol=RTOL)
    st_l5 = oracle_state(sight, num_lines=5)
    got = np.array([R.sample_log_likelihood_k_dlas(st_l5, g["z_kd2"][i], g["nhi_kd2"][i]) for i in range(0, 64, 8)])
    np.testing.assert_allclose(got, g["out_lines5_kd2"][::8], rtol=RTOL)


def test_log_model_evidences_small_seeded():
    """full occupation sweep incl. the legacy-RNG resampling, on the 400-pixel sightline"""
    g = load_golden("evidences_small.npz")
    sight = make_sightline(int(g["sight_index"]), n_unmasked=int(g["sight_n_unmasked"]),
                           n_masked=int(g["sight_n_masked"]), k=int(g["sight_k"]))
    assert checksum(sight.y, sight.this_M, sight.padded_wavelengths, sight.keep) == int(g["sight_crc"])
    st = oracle_state(sight)
    S = int(g["S"])
    from gpy_dla_detection_amd.samples import halton

    nhi = 10.0 ** (20.0 + 3.0 * halton(S, 3))
    np.random.seed(0)
    ev, sll, base = R.log_model_evidences(st, g["dla_sample_z"], nhi, 4, float(g["dla_min_z_separation"]), verbose=False)
    np.testing.assert_array_equal(base, g["dla_base_inds"])
    np.testing.assert_allclose(ev, g["dla_evidences"], rtol=RTOL)
    np.testing.assert_allclose(sll, g["dla_sample_ll"], rtol=RTOL, equal_nan=True)
    assert np.array_equal(np.isnan(sll), np.isnan(g["dla_sample_ll"]))

    nhi_sub = 10.0 ** (19.5 + 0.5 * halton(S, 3))
    np.random.seed(0)
    ev, sll, _ = R.log_model_evidences(st, g["dla_sample_z"], nhi_sub, 1, float(g["dla_min_z_separation"]), verbose=False)
    np.testing.assert_allclose(ev, g["sub_evidences"], rtol=RTOL)
    np.testing.assert_allclose(sll, g["sub_sample_ll"], rtol=RTOL)


def test_oracle_reproduces_reference_log_posterior():
    """G9: the MCMC target density = log prior + the same likelihood (reference log_posterior_mcmc.py:46-95)"""
    from tests.golden_inputs import mcmc_pdf, mcmc_thetas

    g = load_golden("log_posterior.npz")
    sight = make_sightline(1, n_unmasked=400, n_masked=9)
    st = oracle_state(sight)
    z_lo, z_hi = sight.z_range()
    thetas = mcmc_thetas(z_lo, z_hi)
    for theta, want in zip(thetas, g["log_posterior"]):
        z, log_nhi = theta
        inside = z_lo < z < z_hi and 20.0 < log_nhi < 23.0
        if not inside:
            assert np.isneginf(want)
            continue
        got = np.log(mcmc_pdf(log_nhi)) + R.sample_log_likelihood_k_dlas(st, [z], [10.0 ** log_nhi])
        assert abs(got - want) <= 1e-11 * abs(want)


def test_oracle_reproduces_reference_voigt_lls():
    """G10: Lyman-series lines + Lyman-limit break (reference voigt_lls.py:258-360)"""
    g = load_golden("voigt_lls.npz")
    for broadening in (True, False):
        want = g["absorption_b%d" % broadening]
        for i, (z, ln) in enumerate(zip(g["z"], g["log_nhi"])):
            got = R.voigt_absorption(g["grid"], nhi=10.0 ** ln, z_dla=z, num_lines=3, broadening=broadening,
                                     lls_break=True)
            np.testing.assert_allclose(got, want[i], rtol=1e-13, atol=1e-300)
    assert (g["absorption_b0"] < 1e-3).any() and (g["absorption_b0"] > 0.5).any()  # the break is on the grid


def test_oracle_reproduces_reference_on_gapped_masks():
    """G11: sightlines whose masks have long masked runs (reference dla_gp.py:360-388 keeps `~pixel_mask[ind_unmasked]`)"""
    from tests.golden_inputs import GAPPED_CASES, checksum, gapped_sightline

    g = load_golden("gapped.npz")
    for c, case in enumerate(GAPPED_CASES):
        sight = gapped_sightline(**case)
        assert checksum(sight.y, sight.this_M, sight.padded_wavelengths, sight.keep) == int(g["crc_%d" % c])
        st = oracle_state(sight)
        got = np.array([R.sample_log_likelihood_k_dlas(st, [z], [n]) for z, n in zip(g["z1_%d" % c], g["n1_%d" % c])])
        np.testing.assert_allclose(got, g["out1_%d" % c], rtol=1e-12)
        got2 = np.array([R.sample_log_likelihood_k_dlas(st, z, n) for z, n in zip(g["z2_%d" % c], g["n2_%d" % c])])
        np.testing.assert_allclose(got2, g["out2_%d" % c], rtol=1e-12)
        assert abs(R.null_log_model_evidence(st) - g["null_%d" % c]) <= 1e-12 * abs(g["null_%d" % c])


def test_oracle_fan_out_equals_the_oracle_in_process():
    """tests/oracle_fanout.py (the GPU tests' whole-population checker): same values as the oracle called here"""
    from gpy_dla_detection_amd.synthetic import make_qmc_samples, make_sightline
    from tests.oracle_fanout import oracle_log_likelihoods

    jobs = []
    for index, kd in ((2, 1), (5, 3)):
        sight = make_sightline(index, n_unmasked=300, n_masked=9, k=20)
        z, _, nhi = make_qmc_samples(sight, 23)
        zs = np.stack([np.roll(z, j) for j in range(kd)], axis=1)
        ns = np.stack([np.roll(nhi, j) for j in range(kd)], axis=1)
        jobs.append((sight, zs if kd > 1 else z, ns if kd > 1 else nhi))
    got = oracle_log_likelihoods(jobs, workers=3)
    for (sight, zs, ns), values in zip(jobs, got):
        st = oracle_state(sight)
        zs, ns = np.asarray(zs).reshape(23, -1), np.asarray(ns).reshape(23, -1)
        want = np.array([R.sample_log_likelihood_k_dlas(st, zs[i], ns[i]) for i in range(23)])
        np.testing.assert_allclose(values, want, rtol=1e-13)  # (the workers run one BLAS thread each)
