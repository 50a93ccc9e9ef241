"""
GP model with an intervening sub-DLA (19.5 <= log N_HI < 20), evaluated on the GPU.

Drop-in for the model-evidence surface of the reference class
(gpy_dla_detection/subdla_gp.py:23-346).  The reference duplicates the DLA sweep line for
line with the sub-DLA sample set; here both share `dla_gp._AbsorberGP`.
"""
import numpy as np

from .dla_gp import _AbsorberGP
from .set_parameters import Parameters


class SubDLAGP(_AbsorberGP):
    """
    :param dla_samples: sub-DLA QMC samples (`sample_z_lls`, `nhi_samples`, `Z_dla`, `Z_lls`)
    remaining parameters as in `DLAGP`.
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
        return self.dla_samples.sample_z_lls(self.this_wavelengths, self.z_qso)

    def log_model_evidences(self, max_dlas: int = 1) -> np.ndarray:
        """
        [log p(D | 1 sub-DLA), ...] (reference subdla_gp.py:90-222).  Stores
        `sample_log_likelihoods`; like the reference it does not keep `base_sample_inds`.
        """
        evidences, self.sample_log_likelihoods, _ = self._sweep(max_dlas)
        return evidences

    def log_priors(self, z_qso: float, max_dlas: int) -> np.ndarray:
        """
        DLA occupation prior rescaled by the ratio of the column-density partition functions,
        P(>= k sub-DLAs) = Z_lls / Z_dla (M/N)^k (reference subdla_gp.py:311-346).
        """
        num_dlas, num_quasars = self.prior.less_ind(z_qso)
        samples = self.dla_samples
        z_lls = getattr(samples, "_Z_lls", None)
        z_dla = getattr(samples, "_Z_dla", None)
        if z_lls is None:
            z_lls, z_dla = samples.Z_lls, samples.Z_dla
        at_least = z_lls / z_dla * (num_dlas / num_quasars) ** np.arange(1, max_dlas + 1)
        at_least[:-1] = at_least[:-1] - at_least[1:]
        return np.log(at_least)

    @classmethod
    def from_mat(cls, params: Parameters, prior, dla_samples, min_z_separation: float = 3000.0,
                 learned_file: str = "learned_qso_model_lyseries_variance_kim_dr9q_minus_concordance.mat",
                 broadening: bool = True, **kwargs):
        """Load the learned model from the MATLAB v7.3 file (reference SubDLAGPMAT, :349-391)."""
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


def SubDLAGPMAT(params, prior, dla_samples, min_z_separation: float = 3000.0,
                learned_file: str = "learned_qso_model_lyseries_variance_kim_dr9q_minus_concordance.mat",
                broadening: bool = True):
    return SubDLAGP.from_mat(params, prior, dla_samples, min_z_separation, learned_file, broadening)
