"""
GP model with k intervening DLAs; the QMC model-evidence sweep runs on the GPU.

Drop-in for the model-evidence surface of the reference class
(gpy_dla_detection/dla_gp.py:25-472): `log_model_evidences`, `sample_log_likelihood_k_dlas`,
`this_dla_gp`, `log_priors`, `maximum_a_posteriori`, and the side effects callers read
(`sample_log_likelihoods`, `base_sample_inds`).

What moved to the device: the whole occupation sweep (reference :92-225).  The body of the loop over
the QMC samples (:132-159) -- all `num_dla_samples` Voigt-absorbed models and low-rank Gaussian
log-densities of a level -- is one launch sequence, and the levels are chained on the device
(separation masking, log-mean-exp evidence, posterior resampling, :164-218; `dlagp_batch_sweep`).
The resampling takes its random numbers from the legacy global NumPy RNG exactly as
`np.random.choice` would (one `random_sample(S)` per resampled level), so `np.random.seed`
reproduces the reference's `base_sample_inds` and leaves the generator in the reference's state.
`host_levels = True` on an instance (or the class) keeps the levels' host half in NumPy.
"""
from typing import Tuple

import numpy as np

from .null_gp import NullGP
from .set_parameters import Parameters


class _AbsorberGP(NullGP):
    """machinery shared by the DLA and sub-DLA models (the reference keeps two copies of it)"""

    def _init_absorber(self, dla_samples, min_z_separation: float, broadening: bool) -> None:
        self.min_z_separation = self.params.kms_to_z(min_z_separation)
        self.dla_samples = dla_samples
        self.broadening = broadening

    def _sample_redshifts(self) -> np.ndarray:
        raise NotImplementedError

    # -- one sample ----------------------------------------------------------------------------
    def this_dla_gp(self, z_dlas: np.ndarray, nhis: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """
        (dla_mu, dla_M, dla_omega2): the GP with the given absorbers imprinted
        (reference dla_gp.py:331-396).  The absorption vector comes from the device
        (dlagp_absorption); the three products are formed on the host because callers
        (plotting) want NumPy arrays.
        """
        assert len(z_dlas) == len(nhis)
        absorption = self.device_sightline().absorption(z_dlas, nhis)
        assert len(absorption) == len(self.this_mu)
        return (
            self.this_mu * absorption,
            self.this_M * absorption[:, None],
            self.this_omega2 * absorption ** 2,
        )

    def sample_log_likelihood_k_dlas(self, z_dlas: np.ndarray, nhis: np.ndarray) -> float:
        """log p(y | {z_dla, N_HI}_1..k) for one parameter set (reference dla_gp.py:311-329)"""
        assert len(z_dlas) == len(nhis)
        if not self._stock_absorption():
            dla_mu, dla_M, dla_omega2 = self.this_dla_gp(z_dlas, nhis)
            return self.log_mvnpdf_low_rank(self.y, dla_mu, dla_M, dla_omega2 + self.v)
        z = np.asarray(z_dlas, dtype=np.float64)[None, :]
        n = np.asarray(nhis, dtype=np.float64)[None, :]
        return float(self.device_sightline().eval(z, n)[0])

    def _stock_absorption(self) -> bool:
        # a subclass that overrides this_dla_gp (e.g. the LLS variant, reference
        # examples/gp_find_lls.py:159-224) changes the model: honour it, one sample at a time
        return type(self).this_dla_gp is _AbsorberGP.this_dla_gp

    def _level_log_likelihoods(self, z: np.ndarray, nhi: np.ndarray) -> np.ndarray:
        """(S, kd) parameter sets -> (S,) log-likelihoods; one device launch"""
        if self._stock_absorption():
            return self.device_sightline().eval(z, nhi)
        return np.array([self.sample_log_likelihood_k_dlas(z[i], nhi[i]) for i in range(z.shape[0])])

    # -- the sweep -----------------------------------------------------------------------------
    #: True: the levels' host half (mask, evidence, np.random.choice) runs in NumPy exactly as in the
    #: reference.  False (default): all levels are chained on the device (dlagp_batch_sweep); the legacy
    #: global RNG is consumed exactly as `np.random.choice` would consume it -- one `random_sample(S)` per
    #: resampled level -- so `np.random.seed` gives the reference's `base_sample_inds` and leaves the
    #: same RNG state behind.
    host_levels = False

    def _sweep(self, max_dlas: int):
        sample_z = np.asarray(self._sample_redshifts(), dtype=np.float64)
        sample_nhi = np.asarray(self.dla_samples.nhi_samples, dtype=np.float64)
        S = self.params.num_dla_samples
        if self.host_levels or not self._stock_absorption() or max_dlas > 8:
            return occupation_sweep(self._level_log_likelihoods, sample_z, sample_nhi, max_dlas,
                                    self.min_z_separation, num_samples=S)
        return self._sweep_on_device(sample_z, sample_nhi, max_dlas, S)

    def _sweep_on_device(self, sample_z: np.ndarray, sample_nhi: np.ndarray, max_dlas: int, S: int):
        # np.random.choice(S, S, p=...) draws random_sample(S) once per call (reference dla_gp.py:213-218):
        # draw what a full sweep would draw, and give back what an early stop would not have drawn
        rng_before = np.random.get_state()
        uniforms = np.random.random_sample((1, max(max_dlas - 1, 0), S))
        batch, own = self.device_batch()  # the spectrum's shared sightline keeps ONE batch for all its sweeps
        try:
            evidences, sample_ll, base = batch.sweep(sample_z[None, :], sample_nhi, max_dlas,
                                                     self.min_z_separation, uniforms)
        finally:
            if own:
                batch.close()
        evidences, sample_ll, base = evidences[0], sample_ll[0], base[0]
        settle_rng_after_sweep(evidences, max_dlas, S, rng_before)
        return evidences, sample_ll, base


def settle_rng_after_sweep(evidences: np.ndarray, max_dlas: int, S: int, rng_before) -> None:
    """
    The device sweep was handed the uniforms of a FULL sweep; the reference stops at the first level whose evidence
    is NaN (dla_gp.py:200-206) and has then drawn only the levels before it.  Prints the reference's message and
    leaves the legacy global RNG where the reference would have left it.
    """
    nan_levels = np.flatnonzero(np.isnan(evidences))
    if nan_levels.size and nan_levels[0] + 1 < max_dlas:
        stopped = int(nan_levels[0])
        print("Finish the loop earlier because NaN value in log p(D | z_QSO, {} DLAs)".format(stopped))
        np.random.set_state(rng_before)
        for _ in range(stopped):  # the levels before it did resample
            np.random.random_sample(S)


def level_parameters(sample_z: np.ndarray, sample_nhi: np.ndarray, base_sample_inds: np.ndarray, level: int):
    """
    (z, nhi), each (S, level + 1): the i-th QMC point plus `level` points drawn from the previous
    levels' posteriors (reference dla_gp.py:135-150).  Also returns the (level + 1, S) index table.
    """
    S = sample_z.shape[0]
    idx = np.concatenate([np.arange(S)[None, :], base_sample_inds[:level, :]], axis=0)
    return np.ascontiguousarray(sample_z[idx].T), np.ascontiguousarray(sample_nhi[idx].T), idx


def finish_level(sample_log_likelihoods: np.ndarray, level: int, sample_z: np.ndarray, idx: np.ndarray,
                 min_z_separation: float, max_dlas: int, base_sample_inds: np.ndarray):
    """
    Host half of one occupation level (reference dla_gp.py:164-218), given the level's column of
    `sample_log_likelihoods` already holding `log-likelihood - log S`:
    separation masking, log-mean-exp evidence, and -- unless this is the last level or the evidence is
    NaN -- the posterior resampling of base indices with the legacy global NumPy RNG.

    :return: (evidence, keep_going)
    """
    S = sample_z.shape[0]
    if level > 0:  # reference :164-177
        too_close = np.any(np.diff(np.sort(sample_z[idx], axis=0), axis=0) < min_z_separation, axis=0)
        sample_log_likelihoods[too_close, level] = np.nan

    top = np.nanmax(sample_log_likelihoods[:, level])
    probabilities = np.exp(sample_log_likelihoods[:, level] - top)
    evidence = top + np.log(np.nanmean(probabilities)) - np.log(S) * level  # :186-190

    if level + 1 == max_dlas:
        return evidence, False
    if np.isnan(evidence):  # reference :200-206
        print("Finish the loop earlier because NaN value in log p(D | z_QSO, {} DLAs)".format(level))
        return evidence, False

    weights = probabilities
    weights[np.isnan(weights)] = 0.0
    base_sample_inds[level, :] = np.random.choice(  # reference :213-218, legacy global RNG
        np.arange(S).astype(np.int32), size=S, replace=True, p=weights / weights.sum()
    )
    return evidence, True


def occupation_sweep(level_log_likelihoods, sample_z: np.ndarray, sample_nhi: np.ndarray, max_dlas: int,
                     min_z_separation: float, num_samples: int):
    """
    The occupation sweep of reference dla_gp.py:92-225 around a batched evaluator
    `level_log_likelihoods(z (S, kd), nhi (S, kd)) -> (S,)`.

    :return: (evidences (max_dlas,), sample_log_likelihoods (S, max_dlas), base_sample_inds (max_dlas - 1, S) int32)
    """
    S = num_samples
    evidences = np.full((max_dlas,), np.nan)
    base_sample_inds = np.zeros((max_dlas - 1, S), dtype=np.int32)
    sample_log_likelihoods = np.full((S, max_dlas), np.nan)
    for level in range(max_dlas):
        z, nhi, idx = level_parameters(sample_z, sample_nhi, base_sample_inds, level)
        sample_log_likelihoods[:, level] = level_log_likelihoods(z, nhi) - np.log(S)  # :155-159
        evidences[level], keep_going = finish_level(
            sample_log_likelihoods, level, sample_z, idx, min_z_separation, max_dlas, base_sample_inds
        )
        if not keep_going:
            break
    return evidences, sample_log_likelihoods, base_sample_inds


def map_traceback(sample_log_likelihoods: np.ndarray, base_sample_inds: np.ndarray, sample_z: np.ndarray,
                  log_nhi: np.ndarray):
    """
    MAP (z_dla, log N_HI) of every occupation level: the best sample of the level and, through the
    resampled base indices, the absorbers it was stacked on (reference dla_gp.py:428-472).
    Shapes (max_dlas, max_dlas), NaN where unused.
    """
    best = np.nanargmax(sample_log_likelihoods, axis=0)
    max_dlas = sample_log_likelihoods.shape[1]
    MAP_z_dla = np.full((max_dlas, max_dlas), np.nan)
    MAP_log_nhi = np.full((max_dlas, max_dlas), np.nan)
    for level, top in enumerate(best):
        members = np.concatenate([[top], base_sample_inds[:level, top]]).astype(int)
        MAP_z_dla[level, : level + 1] = sample_z[members]
        MAP_log_nhi[level, : level + 1] = log_nhi[members]
    return MAP_z_dla, MAP_log_nhi


class DLAGP(_AbsorberGP):
    """
    p(y | lambda, sigma^2, M, omega, c_0, tau_0, beta, tau_kim, beta_kim, {z_dla, log N_HI})

    :param dla_samples: QMC samples (`sample_z_dlas`, `nhi_samples`, `log_nhi_samples`)
    :param min_z_separation: minimum separation of two absorbers, km/s
    :param broadening: apply the SDSS instrumental broadening to the Voigt profile
    remaining parameters as in `NullGP`.
    """

    def __init__(
        self,
        params: Parameters,
        prior,
        dla_samples,
        rest_wavelengths: np.ndarray,
        mu: np.ndarray,
        M: np.ndarray,
        log_omega: np.ndarray,
        log_c_0: float,
        log_tau_0: float,
        log_beta: float,
        prev_tau_0: float = 0.0023,
        prev_beta: float = 3.65,
        min_z_separation: float = 3000.0,
        broadening: bool = True,
        device: int = 0,
    ):
        super().__init__(params, prior, rest_wavelengths, mu, M, log_omega, log_c_0, log_tau_0, log_beta,
                         prev_tau_0, prev_beta, device=device)
        self._init_absorber(dla_samples, min_z_separation, broadening)

    def _sample_redshifts(self) -> np.ndarray:
        return self.dla_samples.sample_z_dlas(self.this_wavelengths, self.z_qso)

    def log_model_evidences(self, max_dlas: int) -> np.ndarray:
        """
        [log p(D | 1 DLA), ..., log p(D | max_dlas DLAs)] by quasi-Monte-Carlo integration
        (reference dla_gp.py:92-225).  Stores `sample_log_likelihoods` (S, max_dlas) and
        `base_sample_inds` (max_dlas - 1, S) for `maximum_a_posteriori`.
        """
        evidences, self.sample_log_likelihoods, self.base_sample_inds = self._sweep(max_dlas)
        return evidences

    def log_priors(self, z_qso: float, max_dlas: int) -> np.ndarray:
        """log P(k DLAs | z_qso), k = 1..max_dlas, with P(>= k) = (M/N)^k (reference :398-426)"""
        num_dlas, num_quasars = self.prior.less_ind(z_qso)
        at_least = (num_dlas / num_quasars) ** np.arange(1, max_dlas + 1)
        at_least[:-1] = at_least[:-1] - at_least[1:]
        return np.log(at_least)

    def maximum_a_posteriori(self):
        """
        MAP (z_dla, log N_HI) of every occupation level, traced back through the resampled
        base indices (reference :428-472).  Shapes (max_dlas, max_dlas), NaN where unused.
        """
        return map_traceback(self.sample_log_likelihoods, self.base_sample_inds, self._sample_redshifts(),
                             self.dla_samples.log_nhi_samples)

    def mcmc_arguments(self) -> tuple:
        """the `args` tuple the reference passes to emcee for `log_posterior` (dla_gp.py:265-287)"""
        from scipy import stats
        from scipy.integrate import quad

        samples = self.dla_samples
        min_z_dla = self.params.min_z_dla(self.this_wavelengths, self.z_qso)
        max_z_dla = self.params.max_z_dla(self.this_wavelengths, self.z_qso)
        uniform = stats.uniform(loc=samples.uniform_min_log_nhi,
                                scale=samples.uniform_max_log_nhi - samples.uniform_min_log_nhi)

        def unnormalized_pdf(nhi):  # polynomial fit of Garnett et al. (2017), reference :249-251
            return np.exp(-1.2695 * nhi ** 2 + 50.863 * nhi - 509.33)

        Z = quad(unnormalized_pdf, samples.fit_min_log_nhi, 25.0)[0]

        def normalized_pdf(nhi):
            return samples.alpha * (unnormalized_pdf(nhi) / Z) + (1 - samples.alpha) * uniform.pdf(nhi)

        return (self.this_wavelengths, self.y, self.v, self.z_qso, min_z_dla, max_z_dla,
                samples.uniform_min_log_nhi, samples.uniform_max_log_nhi, normalized_pdf, self.padded_wavelengths,
                self.this_mu, self.this_M, self.this_omega2, self.pixel_mask, self.ind_unmasked,
                self.params.num_lines)

    def run_mcmc(self, nwalkers: int, kth_dla: int = 1, nsamples: int = 5000, pos=None,
                 skip_initial_state_check: bool = True):
        """
        emcee refinement of the k-th occupation level (reference dla_gp.py:227-309).  The ensemble is
        evaluated through `log_posterior_vectorized`: all walkers of a step in one device launch.
        Needs the optional dependency emcee.
        """
        import emcee  # optional dependency

        from .log_posterior_mcmc import log_posterior_vectorized

        sampler = emcee.EnsembleSampler(nwalkers, kth_dla * 2, log_posterior_vectorized, args=self.mcmc_arguments(),
                                        vectorize=True)
        if pos is None:
            sample_z = self._sample_redshifts()
            pos = np.concatenate(
                [np.random.choice(sample_z, size=(nwalkers, kth_dla)),
                 np.random.choice(self.dla_samples.log_nhi_samples, size=(nwalkers, kth_dla))], axis=1)
        sampler.run_mcmc(pos, nsamples, progress=False, skip_initial_state_check=skip_initial_state_check)
        return sampler

    @classmethod
    def from_mat(cls, params: Parameters, prior, dla_samples, min_z_separation: float = 3000.0,
                 learned_file: str = "learned_qso_model_lyseries_variance_kim_dr9q_minus_concordance.mat",
                 broadening: bool = True, **kwargs):
        """Load the learned model from the MATLAB v7.3 file (reference DLAGPMAT, :475-514)."""
        import h5py  # optional dependency

        with h5py.File(learned_file, "r") as learned:
            args = (
                learned["rest_wavelengths"][:, 0],
                learned["mu"][:, 0],
                learned["M"][()].T,
                learned["log_omega"][:, 0],
                learned["log_c_0"][0, 0],
                learned["log_tau_0"][0, 0],
                learned["log_beta"][0, 0],
            )
        return cls(params, prior, dla_samples, *args, prev_tau_0=0.0023, prev_beta=3.65,
                   min_z_separation=min_z_separation, broadening=broadening, **kwargs)


def DLAGPMAT(params, prior, dla_samples, min_z_separation: float = 3000.0,
             learned_file: str = "learned_qso_model_lyseries_variance_kim_dr9q_minus_concordance.mat",
             broadening: bool = True):
    return DLAGP.from_mat(params, prior, dla_samples, min_z_separation, learned_file, broadening)
