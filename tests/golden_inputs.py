"""
Regenerable inputs shared by tools/gen_golden.py (which runs the reference on them)
and the tests (which run the oracle / the HIP path on them).  Pure data generation.
"""
import zlib

import numpy as np


def checksum(*arrays) -> int:
    """CRC of the raw bytes; lets tests detect drift of regenerated inputs."""
    crc = 0
    for a in arrays:
        crc = zlib.crc32(np.ascontiguousarray(a).tobytes(), crc)
    return crc


MVNPDF_SHAPES = [(2, 3), (7, 1), (50, 5), (333, 20), (1250, 20), (3000, 20), (64, 32)]


def mvnpdf_inputs(idx, n, k):
    """inputs of golden set G3 (random shapes)"""
    rng = np.random.default_rng(1000 + idx)
    M = rng.standard_normal((n, k)) * 0.3
    mu = rng.standard_normal(n)
    d = 0.05 + rng.random(n)
    y = mu + M @ rng.standard_normal(k) + np.sqrt(d) * rng.standard_normal(n)
    return y, mu, M, d


def raw_spectrum_and_model(seed=5, k=20):
    """synthetic raw SDSS-like spectrum + learned model arrays for golden set G7"""
    rng = np.random.default_rng(seed)
    z_qso = 2.9
    obs = 10.0 ** (3.5600 + 1e-4 * np.arange(4400))
    rest = obs / (1 + z_qso)
    flux = 3.0 + 0.5 * np.sin(rest / 40.0) + 0.2 * rng.standard_normal(rest.shape[0])
    noise_variance = 0.04 * (1 + rng.random(rest.shape[0]))
    pixel_mask = rng.random(rest.shape[0]) < 0.03
    rest_grid = np.arange(911.75, 1215.75 + 1e-9, 0.25)
    mu = 1 + 0.2 * np.cos(rest_grid / 30.0)
    M = 0.1 * rng.standard_normal((rest_grid.shape[0], k)) * (0.9 ** np.arange(k))[None, :]
    log_omega = -2.0 + 0.3 * np.sin(rest_grid / 55.0)
    return z_qso, rest, flux, noise_variance, pixel_mask, rest_grid, mu, M, log_omega, -2.3, -5.9, 1.2


def raw_catalogue(num=3, k=20):
    """
    A small synthetic catalogue for golden set G8: the learned model of `raw_spectrum_and_model`
    and `num` raw spectra (observed wavelengths, flux, noise variance, pixel mask) at different
    quasar redshifts, the later ones with one or two damped troughs pressed into the forest.
    """
    _, _, _, _, _, rest_grid, mu, M, log_omega, lc0, lt0, lb = raw_spectrum_and_model(seed=5, k=k)
    spectra, z_qsos = [], []
    obs = 10.0 ** (3.5600 + 1e-4 * np.arange(4400))
    for i in range(num):
        rng = np.random.default_rng(100 + i)
        z_qso = 2.6 + 0.25 * i
        rest = obs / (1 + z_qso)
        flux = 3.0 + 0.5 * np.sin(rest / 40.0)
        for j in range(i):  # i troughs: Lorentzian optical depth around an absorber redshift
            centre = 1215.6701 * (1 + z_qso - 0.35 - 0.3 * j)
            flux = flux * np.exp(-40.0 * (1 + j) / (1.0 + ((obs - centre) / 3.0) ** 2))
        flux = flux + 0.2 * rng.standard_normal(obs.shape[0])
        noise_variance = 0.04 * (1 + rng.random(obs.shape[0]))
        pixel_mask = rng.random(obs.shape[0]) < 0.03
        spectra.append((obs.copy(), flux, noise_variance, pixel_mask))
        z_qsos.append(z_qso)
    return (rest_grid, mu, M, log_omega, lc0, lt0, lb), spectra, z_qsos


def mcmc_pdf(log_nhi):
    """column-density prior density used for golden set G9 (a smooth positive function; our own choice)"""
    return 0.3 * np.exp(-0.5 * ((log_nhi - 20.6) / 0.5) ** 2) + 0.1


def mcmc_thetas(z_lo, z_hi):
    """(z_dla, log N_HI) pairs for G9: inside the prior box, on its edges and outside"""
    rng = np.random.default_rng(99)
    inside = np.stack([z_lo + (z_hi - z_lo) * rng.random(10), 20.0 + 3.0 * rng.random(10)], axis=1)
    outside = np.array([[z_lo - 0.01, 21.0], [z_hi + 0.01, 21.0], [0.5 * (z_lo + z_hi), 19.9],
                        [0.5 * (z_lo + z_hi), 23.1], [z_hi, 21.0]])
    return np.concatenate([inside, outside])


GAPPED_CASES = [  # (index, n_unmasked, masked runs, comb) of golden set G11: masks with long masked runs
    dict(index=4, n_unmasked=600, gaps=((203, 333),), comb=(400, 420, 2)),
    dict(index=5, n_unmasked=600, gaps=((10, 75), (76, 140), (300, 520)), comb=(400, 420, 2)),
    dict(index=6, n_unmasked=900, gaps=((0, 64), (500, 597), (700, 880)), comb=None),
]


def gapped_sightline(index=4, n_unmasked=600, gaps=((203, 333),), comb=(400, 420, 2), k=20):
    """
    A synthetic sightline whose mask has long masked runs (longer than one 64-entry pass of the MFMA kernel) and a
    comb of single pixels -- the shapes real SDSS masks have (reference read_spec.py:38-71: runs of ivar == 0;
    null_gp.py:139-152).
    """
    from gpy_dla_detection_amd.synthetic import make_sightline

    sight = make_sightline(index, n_unmasked=n_unmasked, n_masked=0, k=k)
    keep = sight.keep.copy()
    for lo, hi in gaps:
        keep[lo:hi] = False
    if comb:
        keep[comb[0]:comb[1]:comb[2]] = False
    sel = keep[sight.keep]  # all kept originally: the index map is the identity
    sight.keep = keep
    sight.y, sight.v = sight.y[sel], sight.v[sel]
    sight.this_mu, sight.this_M, sight.this_omega2 = sight.this_mu[sel], sight.this_M[sel], sight.this_omega2[sel]
    return sight


def driver_datasets(max_dlas):
    """
    What the reference driver keeps of one spectrum's model selection (run_bayes_select.py:189-230), as a table:
    dataset name -> function of a context dict(gp, dla_gp, subdla_gp, bayes, wavelengths, z_qso, MAP).  The three per-model
    vectors of `BayesModelSelect` are laid out [null, sub-DLA, DLA(1..max_dlas)].  Used by tools/gen_golden.py (on the
    reference's classes) and by the route-A test (on this package's), so that both store the same things the same way.
    """
    per_model = {"no_dla": lambda v: v[0], "lls": lambda v: v[1], "dla": lambda v: v[-max_dlas:]}
    stored = {
        "min_z_dlas": lambda c: c["dla_gp"].params.min_z_dla(c["wavelengths"], c["z_qso"]),
        "max_z_dlas": lambda c: c["dla_gp"].params.max_z_dla(c["wavelengths"], c["z_qso"]),
        "sample_log_likelihoods_dla": lambda c: c["dla_gp"].sample_log_likelihoods,
        "base_sample_inds": lambda c: c["dla_gp"].base_sample_inds.T,
        "sample_log_likelihoods_lls": lambda c: c["subdla_gp"].sample_log_likelihoods[:, 0],
        "MAP_z_dlas": lambda c: c["MAP"][0],
        "MAP_log_nhis": lambda c: c["MAP"][1],
        "model_posteriors": lambda c: c["bayes"].model_posteriors,
        "p_dlas": lambda c: c["bayes"].p_dla,
        "p_no_dlas": lambda c: c["bayes"].p_no_dla,
    }
    for quantity in ("log_priors", "log_likelihoods", "log_posteriors"):
        for suffix, pick in per_model.items():
            stored["%s_%s" % (quantity, suffix)] = lambda c, q=quantity, pick=pick: pick(getattr(c["bayes"], q))
    return stored


def run_driver_loop(spectra, z_qsos, params, make_models, bayes, max_dlas):
    """
    The reference driver's loop over spectra (run_bayes_select.py:141-230) around ANY implementation of the three GP
    classes: `make_models()` -> (gp, dla_gp, subdla_gp) in the driver's construction order; `set_data` on each, model
    selection over [gp, subdla_gp, dla_gp], MAP; returns the per-quasar datasets stacked.
    """
    import numpy as np

    stored = driver_datasets(max_dlas)
    rows = {name: [] for name in stored}
    for (wavelengths, flux, noise_variance, pixel_mask), z_qso in zip(spectra, z_qsos):
        np.random.seed(0)  # :144
        rest_wavelengths = params.emitted_wavelengths(wavelengths, z_qso)
        gp, dla_gp, subdla_gp = make_models()
        for model in (gp, dla_gp, subdla_gp):
            model.set_data(rest_wavelengths, flux, noise_variance, pixel_mask, z_qso, build_model=True)
        bayes.model_selection([gp, subdla_gp, dla_gp], z_qso)  # the model list of :189
        context = dict(gp=gp, dla_gp=dla_gp, subdla_gp=subdla_gp, bayes=bayes, wavelengths=wavelengths, z_qso=z_qso,
                       MAP=dla_gp.maximum_a_posteriori())
        for name, value_of in stored.items():
            rows[name].append(np.array(value_of(context)))
    return {name: np.stack(values) for name, values in rows.items()}
