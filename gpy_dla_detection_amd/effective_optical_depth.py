"""
Mean Lyman-series forest optical depth per pixel and series member.

Host-side NumPy (runs once per spectrum inside `NullGP.get_interp`; O(n * 31)).
Same definition and return shape as the reference
(gpy_dla_detection/effective_optical_depth.py:10-80):

    tau[p, i] = tau_0 * (f_i lambda_i) / (f_1 lambda_1) * (lambda_obs[p] / lambda_i)^beta
                * [ lambda_obs[p] / lambda_i - 1 <= z_qso ]
"""
import numpy as np

from . import lyman_series


def effective_optical_depth(
    wavelengths: np.ndarray,
    beta: float,
    tau_0: float,
    z_qso: float,
    num_forest_lines: int,
    skip_lya_indicator: bool = True,
) -> np.ndarray:
    """
    :param wavelengths: observed wavelengths (Å), shape (n,)
    :param beta: exponent of the redshift evolution
    :param tau_0: Ly-alpha normalisation
    :param z_qso: quasar redshift; absorbers beyond it contribute nothing
    :param num_forest_lines: number of series members
    :param skip_lya_indicator: accepted for signature compatibility; the
        reference never reads it (the indicator is applied to every member).
    :return: (n, num_forest_lines) optical depths
    """
    wavelengths = np.asarray(wavelengths, dtype=np.float64)
    lines = lyman_series.TRANSITION_WAVELENGTHS[:num_forest_lines] * 1e8  # cm -> Å
    strengths = lyman_series.OSCILLATOR_STRENGTHS[:num_forest_lines]

    # absorber redshift of every pixel with respect to every series member
    z_abs = (wavelengths[:, None] - lines[None, :]) / lines[None, :]

    scale = tau_0 * strengths / strengths[0] * lines / lines[0]
    tau = scale[None, :] * (1 + z_abs) ** beta
    return tau * (z_abs <= z_qso)
