#!/usr/bin/env python3
"""
Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only in the build container, where /root/reference exists.  The reference is
imported read-only (PYTHONDONTWRITEBYTECODE) with two empty stub modules for its
optional third-party imports that this image lacks (`h5py`, `emcee`; they are only
touched by the *.mat loaders and run_mcmc, never by the model-evidence path --
SURVEY.md §8c).  Nothing of the reference's source enters the repository: the
fixtures hold inputs (or the seeds that regenerate them) and the reference's outputs.

    python tools/gen_golden.py            # writes tests/golden/*.npz

Fixtures (all float64):
  G1 voigt_profile.npz     reference voigt.Voigt on a velocity grid x all 31 series members
  G2 voigt_absorption.npz  reference voigt.voigt_absorption on three padded grids
  G3 mvnpdf.npz            NullGP.log_mvnpdf_low_rank: reference test cases + random shapes
  G4 sample_ll.npz         DLAGP.sample_log_likelihood_k_dlas, 1..4 absorbers, synthetic sightline
  G5 evidences_*.npz       DLAGP.log_model_evidences(4) / SubDLAGP.log_model_evidences(1), seed 0
  G6 model_selection.npz   BayesModelSelect([0,1,4],2).model_selection with a fixed prior
  G7 set_data.npz          NullGP.set_data/get_interp on a synthetic raw spectrum + learned model
  G10 voigt_lls.npz        voigt_lls.voigt_absorption: Lyman series + Lyman-limit break
  G9 log_posterior.npz     log_posterior_mcmc.log_posterior (MCMC target density) for a set of thetas
  G8 catalogue.npz         the per-spectrum body of run_bayes_select.process_qso on three raw spectra:
                           every array the reference writes to processed_qsos_multi_meanflux.h5
  G11 gapped.npz           sightlines whose masks have long masked runs: per-sample likelihoods (1, 2 absorbers), null
                           evidence, a seeded DLAGP.log_model_evidences(3)
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)  # repo first: its tests/ package must win over the reference's
for _name in ("h5py", "emcee"):
    sys.modules.setdefault(_name, types.ModuleType(_name))
sys.modules["emcee"].EnsembleSampler = object

import numpy as np  # noqa: E402

from gpy_dla_detection.set_parameters import Parameters as RefParameters  # noqa: E402
from gpy_dla_detection.null_gp import NullGP as RefNullGP  # noqa: E402
from gpy_dla_detection.dla_gp import DLAGP as RefDLAGP  # noqa: E402
from gpy_dla_detection.subdla_gp import SubDLAGP as RefSubDLAGP  # noqa: E402
from gpy_dla_detection.bayesian_model_selection import BayesModelSelect as RefBayes  # noqa: E402
from gpy_dla_detection import voigt as ref_voigt  # noqa: E402
from gpy_dla_detection import log_posterior_mcmc as ref_mcmc  # noqa: E402
from gpy_dla_detection import voigt_lls as ref_voigt_lls  # noqa: E402

from gpy_dla_detection_amd.synthetic import make_sightline, make_qmc_samples  # noqa: E402
from gpy_dla_detection_amd.samples import halton  # noqa: E402
from tests.golden_inputs import (  # noqa: E402
    GAPPED_CASES, MVNPDF_SHAPES, checksum, gapped_sightline, mcmc_pdf, mcmc_thetas, mvnpdf_inputs, raw_catalogue,
    raw_spectrum_and_model,
)

OUT = os.path.join(REPO, "tests", "golden")


class FixedPrior:
    """stands in for PriorCatalog: `less_ind` returns fixed counts (model_priors.py:142-157)"""

    def __init__(self, num_dlas=731.0, num_quasars=9265.0):
        self.counts = (num_dlas, num_quasars)

    def less_ind(self, z_qso):
        return self.counts


class ArraySamples:
    """minimal DLASamplesMAT / SubDLASamplesMAT stand-in built from arrays"""

    def __init__(self, params, offsets, log_nhi):
        self.params = params
        self._off = offsets
        self.log_nhi_samples = log_nhi
        self.nhi_samples = 10.0 ** log_nhi
        self._Z_dla = self.Z_dla = 1.0  # read as private attrs by subdla_gp.py:335-336
        self._Z_lls = self.Z_lls = 0.25

    def sample_z_dlas(self, wavelengths, z_qso):
        lo = self.params.min_z_dla(wavelengths, z_qso)
        hi = self.params.max_z_dla(wavelengths, z_qso)
        return lo + (hi - lo) * self._off

    sample_z_lls = sample_z_dlas


def install_state(gp, sight):
    """put a synthetic sightline's post-set_data state on a reference GP instance"""
    n_u = sight.unmasked_wavelengths.shape[0]
    gp.y = sight.y
    gp.v = sight.v
    gp.this_mu = sight.this_mu
    gp.this_M = sight.this_M
    gp.this_omega2 = sight.this_omega2
    gp.padded_wavelengths = sight.padded_wavelengths
    gp.unmasked_wavelengths = sight.unmasked_wavelengths
    gp.this_wavelengths = sight.this_wavelengths
    gp.ind_unmasked = np.ones(n_u, dtype=bool)
    gp.pixel_mask = ~sight.keep
    gp.z_qso = sight.z_qso
    return gp


def dummy_model(k):
    rest = np.linspace(911.75, 1215.75, 5)
    return rest, np.ones(5), np.ones((5, k)), np.zeros(5), 0.0, 0.0, 0.0


def ref_gp(cls, sight, S, offsets=None, log_nhi=None, **kw):
    params = RefParameters(num_dla_samples=S, k=sight.k)
    args = dummy_model(sight.k)
    if cls is RefNullGP:
        gp = cls(params, FixedPrior(), *args)
    else:
        gp = cls(params, FixedPrior(), ArraySamples(params, offsets, log_nhi), *args, **kw)
    return install_state(gp, sight)


# ----------------------------------------------------------------------------------------------
def g1_voigt_profile():
    x = np.concatenate([[0.0], np.logspace(-3, 4.6, 400)])
    x = np.concatenate([-x[:0:-1], x])  # symmetric, 801 points, dimensionless
    velocity = x * np.sqrt(2) * ref_voigt.sigma
    out = np.stack([ref_voigt.Voigt(velocity, ref_voigt.sigma, g) for g in ref_voigt.gammas])
    np.savez_compressed(os.path.join(OUT, "voigt_profile.npz"), velocity=velocity, profile=out)


def g2_voigt_absorption():
    rng = np.random.default_rng(7)
    grids = {}
    for name, n_u in (("g206", 200), ("g1256", 1250), ("g3067", 3061)):
        lam = 3647.0 * 10.0 ** (1e-4 * np.arange(n_u) * (3061.0 / n_u))
        from gpy_dla_detection_amd.synthetic import pad_wavelengths

        grids[name] = pad_wavelengths(lam)
    cases = []
    store = {}
    for name, grid in grids.items():
        store["grid_" + name] = grid
        z_lo = grid[3] / 1215.6701 - 1
        z_hi = min(grid[-4] / 1215.6701 - 1, 3.0)
        n_pairs = 32 if name == "g206" else 6
        zs = z_lo + (z_hi - z_lo) * rng.random(n_pairs)
        logn = 19.5 + 3.5 * rng.random(n_pairs)
        lines = (1, 3, 5, 31) if name == "g206" else (3, 31)
        for i in range(n_pairs):
            for L in lines:
                for broad in (True, False):
                    prof = ref_voigt.voigt_absorption(grid, 10.0 ** logn[i], zs[i], num_lines=L, broadening=broad)
                    key = "out_%d" % len(cases)
                    store[key] = prof
                    cases.append((name, zs[i], logn[i], L, int(broad)))
    # the two parameter sets of the reference's own tests/test_voigt.py:8-57
    for z_qso, z_dla, logn, L, grid in (
        (3.15, 3.1, 20.3, 3, np.linspace(911, 1216, 1000) * (1 + 3.15)),
        (5.0, 4.5, 21.0, 5, np.linspace(911, 1216) * (1 + 5.0)),
    ):
        name = "lin%d" % grid.shape[0]
        store["grid_" + name] = grid
        for broad in (True, False):
            store["out_%d" % len(cases)] = ref_voigt.voigt_absorption(grid, 10.0 ** logn, z_dla, num_lines=L, broadening=broad)
            cases.append((name, z_dla, logn, L, int(broad)))
    store["case_grid"] = np.array([c[0] for c in cases])
    store["case_z"] = np.array([c[1] for c in cases])
    store["case_log_nhi"] = np.array([c[2] for c in cases])
    store["case_num_lines"] = np.array([c[3] for c in cases])
    store["case_broadening"] = np.array([c[4] for c in cases])
    np.savez_compressed(os.path.join(OUT, "voigt_absorption.npz"), **store)


def g3_mvnpdf():
    store = {}
    # reference tests/test_model.py:52-72 (integer inputs on purpose)
    mu = np.array([1, 2])
    M = np.array([[2, 3, 1], [1, 2, 4]])
    d = np.ones(2) * 2
    ys = np.array([[1, 2], [2, 3], [100, 100]])
    store["kat_y"], store["kat_mu"], store["kat_M"], store["kat_d"] = ys, mu, M, d
    store["kat_out"] = np.array([RefNullGP.log_mvnpdf_low_rank(y, mu, M, d) for y in ys])
    # random shapes, inputs stored for the small ones and regenerated from the seed for the big ones
    shapes = MVNPDF_SHAPES
    outs = []
    sums = []
    for idx, (n, k) in enumerate(shapes):
        y, mu_, M_, d_ = mvnpdf_inputs(idx, n, k)
        outs.append(
            [
                RefNullGP.log_mvnpdf_low_rank(y, mu_, M_, d_, scipy_lapack=True),
                RefNullGP.log_mvnpdf_low_rank(y, mu_, M_, d_, scipy_lapack=False),
            ]
        )
        sums.append(checksum(y, mu_, M_, d_))
    store["rand_shapes"] = np.array(shapes)
    store["rand_out"] = np.array(outs)
    store["rand_input_crc"] = np.array(sums, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "mvnpdf.npz"), **store)


def g4_sample_ll():
    sight = make_sightline(2)  # two injected absorbers, n = 3000, k = 20
    S = 64
    z, log_nhi, nhi = make_qmc_samples(sight, S)
    gp = ref_gp(RefDLAGP, sight, S, halton(S, 2), log_nhi)
    rng = np.random.default_rng(11)
    store = {"sight_index": 2, "sight_crc": checksum(sight.y, sight.this_M, sight.padded_wavelengths, sight.keep)}
    for kd in (1, 2, 3, 4):
        zz = np.empty((S, kd))
        nn = np.empty((S, kd))
        out = np.empty(S)
        for i in range(S):
            others = rng.choice(S, size=kd - 1, replace=False)
            zz[i] = np.concatenate([[z[i]], z[others]])
            nn[i] = np.concatenate([[nhi[i]], nhi[others]])
            out[i] = gp.sample_log_likelihood_k_dlas(zz[i], nn[i])
        store["z_kd%d" % kd], store["nhi_kd%d" % kd], store["out_kd%d" % kd] = zz, nn, out
    # no-broadening and 5-line variants
    gp.broadening = False
    store["out_nobroad_kd1"] = np.array([gp.sample_log_likelihood_k_dlas(store["z_kd1"][i], store["nhi_kd1"][i]) for i in range(S)])
    gp.broadening = True
    gp.params.num_lines = 5
    store["out_lines5_kd2"] = np.array([gp.sample_log_likelihood_k_dlas(store["z_kd2"][i], store["nhi_kd2"][i]) for i in range(S)])
    gp.params.num_lines = 3
    null = ref_gp(RefNullGP, sight, S)
    store["null_evidence"] = null.log_model_evidence()
    np.savez_compressed(os.path.join(OUT, "sample_ll.npz"), **store)


def g11_gapped():
    store = {}
    for c, case in enumerate(GAPPED_CASES):
        sight = gapped_sightline(**case)
        store["crc_%d" % c] = checksum(sight.y, sight.this_M, sight.padded_wavelengths, sight.keep)
        S = 40
        z, log_nhi, nhi = make_qmc_samples(sight, S)
        gp = ref_gp(RefDLAGP, sight, S, halton(S, 2), log_nhi)
        store["z1_%d" % c], store["n1_%d" % c] = z, nhi
        store["out1_%d" % c] = np.array([gp.sample_log_likelihood_k_dlas(z[i:i + 1], nhi[i:i + 1]) for i in range(S)])
        z2, n2 = np.stack([z[:16], z[::-1][:16]], axis=1), np.stack([nhi[:16], nhi[::-1][:16]], axis=1)
        store["z2_%d" % c], store["n2_%d" % c] = z2, n2
        store["out2_%d" % c] = np.array([gp.sample_log_likelihood_k_dlas(z2[i], n2[i]) for i in range(16)])
        store["null_%d" % c] = ref_gp(RefNullGP, sight, S).log_model_evidence()
        S = 64
        off, u = halton(S, 2), halton(S, 3)
        dla = ref_gp(RefDLAGP, sight, S, off, 20.0 + 3.0 * u, min_z_separation=3000.0)
        np.random.seed(0)
        store["sweep_ev_%d" % c] = dla.log_model_evidences(3)
        store["sweep_ll_%d" % c] = dla.sample_log_likelihoods
        store["sweep_base_%d" % c] = dla.base_sample_inds
        store["sweep_z_%d" % c] = dla.dla_samples.sample_z_dlas(dla.this_wavelengths, dla.z_qso)
        store["sweep_nhi_%d" % c] = dla.dla_samples.nhi_samples
        store["min_z_separation"] = dla.min_z_separation
    np.savez_compressed(os.path.join(OUT, "gapped.npz"), **store)


def g5_evidences():
    for tag, sight_kwargs, S in (
        ("small", dict(index=1, n_unmasked=400, n_masked=9, k=20), 256),
        ("full", dict(index=2), 256),
        ("full2k", dict(index=1), 2048),
    ):
        sight = make_sightline(**sight_kwargs)
        store = {"sight_crc": checksum(sight.y, sight.this_M, sight.padded_wavelengths, sight.keep), "S": S}
        for k_, v_ in sight_kwargs.items():
            store["sight_" + k_] = v_
        off = halton(S, 2)
        u = halton(S, 3)
        dla = ref_gp(RefDLAGP, sight, S, off, 20.0 + 3.0 * u, min_z_separation=3000.0)
        np.random.seed(0)
        store["dla_evidences"] = dla.log_model_evidences(4)
        store["dla_sample_ll"] = dla.sample_log_likelihoods
        store["dla_base_inds"] = dla.base_sample_inds
        store["dla_sample_z"] = dla.dla_samples.sample_z_dlas(dla.this_wavelengths, dla.z_qso)
        store["dla_min_z_separation"] = dla.min_z_separation
        MAP_z, MAP_l = dla.maximum_a_posteriori()
        store["dla_map_z"], store["dla_map_log_nhi"] = MAP_z, MAP_l
        sub = ref_gp(RefSubDLAGP, sight, S, off, 19.5 + 0.5 * u, min_z_separation=3000.0)
        np.random.seed(0)
        store["sub_evidences"] = sub.log_model_evidences(1)
        store["sub_sample_ll"] = sub.sample_log_likelihoods
        null = ref_gp(RefNullGP, sight, S)
        store["null_evidence"] = null.log_model_evidence()
        np.savez_compressed(os.path.join(OUT, "evidences_%s.npz" % tag), **store)


def g6_model_selection():
    sight = make_sightline(1, n_unmasked=400, n_masked=9)
    S = 128
    off = halton(S, 2)
    u = halton(S, 3)
    null = ref_gp(RefNullGP, sight, S)
    sub = ref_gp(RefSubDLAGP, sight, S, off, 19.5 + 0.5 * u)
    dla = ref_gp(RefDLAGP, sight, S, off, 20.0 + 3.0 * u)
    bayes = RefBayes([0, 1, 4], 2)
    np.random.seed(0)
    log_post = bayes.model_selection([null, sub, dla], sight.z_qso)
    np.savez_compressed(
        os.path.join(OUT, "model_selection.npz"),
        S=S,
        log_posteriors=log_post,
        log_priors=bayes.log_priors,
        log_likelihoods=bayes.log_likelihoods,
        model_posteriors=bayes.model_posteriors,
        p_dla=bayes.p_dla,
        sight_crc=checksum(sight.y, sight.this_M),
    )


def g7_set_data():
    z_qso, rest, flux, nv, mask, rest_grid, mu, M, log_omega, lc0, lt0, lb = raw_spectrum_and_model()
    params = RefParameters()
    gp = RefNullGP(params, FixedPrior(), rest_grid, mu, M, log_omega, lc0, lt0, lb)
    gp.set_data(rest, flux, nv, mask, z_qso, normalize=True, build_model=True)
    np.savez_compressed(
        os.path.join(OUT, "set_data.npz"),
        x=gp.x,
        y=gp.y,
        v=gp.v,
        this_mu=gp.this_mu,
        this_M=gp.this_M,
        this_omega2=gp.this_omega2,
        padded_wavelengths=gp.padded_wavelengths,
        unmasked_wavelengths=gp.unmasked_wavelengths,
        this_wavelengths=gp.this_wavelengths,
        ind_unmasked=gp.ind_unmasked,
        normalization_median=gp.normalization_median,
        null_evidence=gp.log_model_evidence(),
        min_z_dla=params.min_z_dla(gp.this_wavelengths, z_qso),
        max_z_dla=params.max_z_dla(gp.this_wavelengths, z_qso),
    )


def g8_catalogue():
    """
    The loop body of the reference driver (run_bayes_select.py:141-245) with the reference's classes,
    on raw_catalogue(): read_spec is replaced by the synthetic spectra, the *.mat loaders by arrays.
    """
    S, max_dlas = 96, 3
    learned, spectra, z_qsos = raw_catalogue()
    params = RefParameters(num_dla_samples=S)
    prior = FixedPrior()
    off, u = halton(S, 2), halton(S, 3)
    dla_samples = ArraySamples(params, off, 20.0 + 3.0 * u)
    sub_samples = ArraySamples(params, off, 19.5 + 0.5 * u)
    bayes = RefBayes([0, 1, max_dlas], 2)
    from tests.golden_inputs import run_driver_loop

    # the driver's loop (run_bayes_select.py:141-230) around the REFERENCE's classes: its per-quasar datasets, stacked
    out = dict(run_driver_loop(
        spectra, z_qsos, params,
        lambda: (RefNullGP(params, prior, *learned),
                 RefDLAGP(params, prior, dla_samples, *learned, min_z_separation=3000.0, broadening=True),
                 RefSubDLAGP(params, prior, sub_samples, *learned, min_z_separation=3000.0, broadening=True)),
        bayes, max_dlas))
    out["z_qsos"] = np.array(z_qsos)
    out["S"], out["max_dlas"] = S, max_dlas
    out["input_crc"] = checksum(*[a for spec in spectra for a in spec[:3]])
    np.savez_compressed(os.path.join(OUT, "catalogue.npz"), **out)


def g9_log_posterior():
    """reference log_posterior_mcmc.log_posterior on a synthetic sightline, incl. thetas outside the prior box"""
    sight = make_sightline(1, n_unmasked=400, n_masked=9)
    gp = ref_gp(RefNullGP, sight, 16)
    z_lo, z_hi = sight.z_range()
    thetas = mcmc_thetas(z_lo, z_hi)
    args = (gp.this_wavelengths, gp.y, gp.v, gp.z_qso, z_lo, z_hi, 20.0, 23.0, mcmc_pdf, gp.padded_wavelengths,
            gp.this_mu, gp.this_M, gp.this_omega2, gp.pixel_mask, gp.ind_unmasked, 3)
    values = np.array([ref_mcmc.log_posterior(tuple(t), *args) for t in thetas])
    np.savez_compressed(os.path.join(OUT, "log_posterior.npz"), thetas=thetas, log_posterior=values,
                        sight_crc=checksum(sight.y, sight.this_M))


def g10_voigt_lls():
    """reference voigt_lls.voigt_absorption (Lyman series + Lyman-limit break) on a padded grid"""
    grid = 10.0 ** (3.5600 + 1e-4 * (np.arange(1256) - 3))  # 3629-4845 A: the limit of z = 3.1..4.2 absorbers is inside
    rng = np.random.default_rng(17)
    z = 3.05 + 1.2 * rng.random(12)
    log_nhi = 17.0 + 5.5 * rng.random(12)
    store = {"grid": grid, "z": z, "log_nhi": log_nhi}
    for broadening in (True, False):
        store["absorption_b%d" % broadening] = np.stack([
            ref_voigt_lls.voigt_absorption(grid, nhi=10.0 ** ln, z_lls=zz, num_lines=3, broadening=broadening)
            for zz, ln in zip(z, log_nhi)
        ])
    np.savez_compressed(os.path.join(OUT, "voigt_lls.npz"), **store)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11"]
    table = dict(g1=g1_voigt_profile, g2=g2_voigt_absorption, g3=g3_mvnpdf, g4=g4_sample_ll,
                 g5=g5_evidences, g6=g6_model_selection, g7=g7_set_data, g8=g8_catalogue, g9=g9_log_posterior, g10=g10_voigt_lls, g11=g11_gapped)
    for name in which:
        print("generating", name, flush=True)
        table[name]()
    for f in sorted(os.listdir(OUT)):
        print("%8d  %s" % (os.path.getsize(os.path.join(OUT, f)), f))
