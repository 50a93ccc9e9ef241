"""
Bayesian model selection over [null, sub-DLA, DLA(1..k)] -- the consumer of the GPU sweep.

Host-side: six numbers per spectrum.  Same surface as the reference class
(gpy_dla_detection/bayesian_model_selection.py:20-149): `model_selection(model_list, z_qso)`
returns the log posteriors and leaves `log_priors`, `log_likelihoods`, `log_posteriors`
on the instance; `model_posteriors`, `p_dla`, `p_no_dla` derive from them.
"""
from typing import List

import numpy as np
from scipy.special import logsumexp as _scipy_logsumexp

from .dla_gp import DLAGP, _AbsorberGP, settle_rng_after_sweep
from .null_gp import NullGP
from .subdla_gp import SubDLAGP


def logsumexp(a) -> float:
    """
    `scipy.special.logsumexp` of a short 1-d array (the reference's calls, bayesian_model_selection.py:80,132-149).
    scipy's general-purpose routine costs 60 us a call -- three calls per spectrum in the reference's driver loop; for
    finite input the textbook form agrees with it to the last bit or two, anything else (the NaN of a stopped sweep,
    infinities) goes to scipy.
    """
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1 and a.size and np.isfinite(a).all():
        top = a.max()
        return top + np.log(np.exp(a - top).sum())
    return _scipy_logsumexp(a)


class BayesModelSelect:
    """
    :param all_max_dlas: absorbers to integrate per model of the list; 0 marks the null model
    :param dla_model_ind: position of the DLA model in the list
    """

    def __init__(self, all_max_dlas: List[int] = [0, 1, 4], dla_model_ind: int = 2):
        self.all_max_dlas = all_max_dlas
        self.dla_model_ind = dla_model_ind

    def model_selection(self, model_list: List[NullGP], z_qso: float) -> np.ndarray:
        """
        log p(M_i | D) + const for every (model, occupation) pair, in list order
        (reference :48-109).  The first model is the null model, the last the DLA model.
        """
        assert isinstance(model_list[-1], DLAGP)
        assert isinstance(model_list[-1], NullGP)
        assert len(model_list) > self.dla_model_ind

        # The reference's model list on one shared sightline: the device starts on the sub-DLA sweep (no random numbers
        # needed) while the priors are assembled and the occupation sweep's uniforms are drawn here on the host.  Neither
        # touches what the other reads, so the order of the two is free.
        finish_evidences = self._start_evidences_in_one_call(model_list)

        # priors of the absorber models; the null model takes what is left (reference :68-80)
        try:
            priors = []
            for model, num in zip(model_list, self.all_max_dlas):
                priors.extend([np.nan] if num == 0 else list(model.log_priors(z_qso, num)))
            log_priors = np.array(priors)
            log_priors[0] = np.log(1 - np.exp(logsumexp(log_priors[1:])))
        except BaseException:
            # the call started above will never be finished: the spectrum's shared batch must not keep it pending
            # (the GP objects go on using that batch)
            if finish_evidences is not None:
                finish_evidences.cancel()
            raise

        # evidences: this is where the GPU sweep runs (reference :84-96)
        log_likelihoods = finish_evidences() if finish_evidences is not None else None
        if log_likelihoods is None:
            evidences = []
            for model, num in zip(model_list, self.all_max_dlas):
                if num == 0:
                    evidences.append(model.log_model_evidence())
                else:
                    evidences.extend(model.log_model_evidences(num))
            log_likelihoods = np.array(evidences)

        self.log_priors = log_priors
        self.log_likelihoods = log_likelihoods
        self.log_posteriors = log_likelihoods + log_priors
        return self.log_posteriors

    def _start_evidences_in_one_call(self, model_list: List[NullGP]):
        """
        The reference driver's model list [null, sub-DLA (1), DLA (1..k)] on ONE spectrum (run_bayes_select.py:189):
        when the three stock models share one HBM-resident sightline (`null_gp.SharedSightline` -- equal learned
        model, settings and spectrum by content), the loop above is one call into the library
        (dlagp_batch_model_evidences: null evidence, sub-DLA sweep and occupation sweep enqueued back to back, one
        wait).  Same numbers bit for bit, same side effects on the models (`sample_log_likelihoods`,
        `base_sample_inds`), same use of the legacy global RNG.  Returns None when the shape does not apply, else -- with
        the first phase (uploads, sub-DLA sweep) already running on the device -- the function that draws the uniforms,
        finishes the call and returns the log-likelihoods.
        """
        if len(model_list) != 3 or list(self.all_max_dlas[:2]) != [0, 1] or len(self.all_max_dlas) != 3:
            return None
        null, sub, dla = model_list
        max_dlas = self.all_max_dlas[2]
        if not (type(null).log_model_evidence is NullGP.log_model_evidence and isinstance(sub, SubDLAGP)
                and type(sub).log_model_evidences is SubDLAGP.log_model_evidences
                and type(dla).log_model_evidences is DLAGP.log_model_evidences and 1 <= max_dlas <= 8):
            return None
        for model in (sub, dla):
            if model.host_levels or not model._stock_absorption() or type(model)._sweep is not _AbsorberGP._sweep:
                return None
        S = dla.params.num_dla_samples
        if sub.params.num_dla_samples != S or sub.min_z_separation != dla.min_z_separation:
            return None
        try:
            devs = [m.device_sightline() for m in model_list]
        except AttributeError:  # a model without data: let the plain loop raise what the reference raises
            return None
        shared = dla._shared
        if shared is None or any(d is not devs[0] for d in devs) or sub._shared is not shared or null._shared is not shared:
            return None
        z_lls = np.asarray(sub._sample_redshifts(), dtype=np.float64)
        z_dla = np.asarray(dla._sample_redshifts(), dtype=np.float64)
        nhi_sub = np.asarray(sub.dla_samples.nhi_samples, dtype=np.float64)
        nhi_dla = np.asarray(dla.dla_samples.nhi_samples, dtype=np.float64)
        if not (z_lls.shape == z_dla.shape == nhi_sub.shape == nhi_dla.shape == (S,)):
            return None
        batch = shared.batch
        batch.model_evidences_start(z_lls[None, :], nhi_sub, z_dla[None, :], nhi_dla, max_dlas, dla.min_z_separation)

        def finish():
            # the sub-DLA sweep (one level) draws nothing; the occupation sweep draws one random_sample(S) per resampled
            # level, exactly as in DLAGP.log_model_evidences
            rng_before = np.random.get_state()
            uniforms = np.random.random_sample((1, max(max_dlas - 1, 0), S))
            nulls, (ev_sub, ll_sub, _), (ev_dla, ll_dla, base) = batch.model_evidences_finish(uniforms)
            sub.sample_log_likelihoods = ll_sub[0]
            dla.sample_log_likelihoods, dla.base_sample_inds = ll_dla[0], base[0]
            settle_rng_after_sweep(ev_dla[0], max_dlas, S, rng_before)
            return np.concatenate([nulls, ev_sub[0], ev_dla[0]])

        finish.cancel = batch.model_evidences_cancel
        return finish

    @property
    def dla_model_posterior_ind(self) -> np.ndarray:
        """mask of the DLA(1..k) entries of `log_posteriors` (reference :111-124)"""
        ind = np.zeros((self.log_posteriors.shape[0],), dtype=np.bool_)
        ind[-self.all_max_dlas[self.dla_model_ind]:] = True
        self._dla_model_posterior_ind = ind
        return ind

    @staticmethod
    def _normalised(log_values: np.ndarray) -> np.ndarray:
        return np.exp(log_values - logsumexp(log_values))

    @property
    def model_posteriors(self) -> np.ndarray:
        return self._normalised(self.log_posteriors)

    @property
    def model_evidences(self) -> np.ndarray:
        return self._normalised(self.log_likelihoods)

    @property
    def model_priors(self) -> np.ndarray:
        return self._normalised(self.log_priors)

    @property
    def p_dla(self) -> float:
        self._p_dla = np.sum(self.model_posteriors[self.dla_model_posterior_ind])
        return self._p_dla

    @property
    def p_no_dla(self) -> float:
        return 1 - self.p_dla
