"""
Bayesian model selection over [null, sub-DLA, DLA(1..k)] -- the consumer of the GPU sweep.

Host-side: six numbers per spectrum.  Same surface as the reference class
(gpy_dla_detection/bayesian_model_selection.py:20-149): `model_selection(model_list, z_qso)`
returns the log posteriors and leaves `log_priors`, `log_likelihoods`, `log_posteriors`
on the instance; `model_posteriors`, `p_dla`, `p_no_dla` derive from them.
"""
from typing import List

import numpy as np
from scipy.special import logsumexp

from .dla_gp import DLAGP
from .null_gp import NullGP


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

        # priors of the absorber models; the null model takes what is left (reference :68-80)
        priors = []
        for model, num in zip(model_list, self.all_max_dlas):
            priors.extend([np.nan] if num == 0 else list(model.log_priors(z_qso, num)))
        log_priors = np.array(priors)
        log_priors[0] = np.log(1 - np.exp(logsumexp(log_priors[1:])))

        # evidences: this is where the GPU sweep runs (reference :84-96)
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
