"""
Deterministic synthetic sightlines at the kernel's input boundary.

No learned model, spectra or sample files ship with the reference, so tests and
bench.py run on synthetic post-`set_data` state: flux, noise, interpolated GP
mean / low-rank factor / absorption noise, the padded observed-wavelength grid
and the pixel mask -- the arrays `NullGP.set_data` leaves on the instance
(reference gpy_dla_detection/null_gp.py:117-177, :240-242).

This module only *generates data*.  The injected absorbers use SciPy's Faddeeva
function on the host; nothing here is on the evaluated hot path.
"""
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np
from scipy.special import wofz

from . import lyman_series as ls
from .samples import halton

LYA_ANGSTROM = 1215.6701
BASE_SEED = 20241024


@dataclass
class SyntheticSightline:
    """Post-`set_data` state of one sightline (float64 everywhere, `keep` bool)."""

    z_qso: float
    y: np.ndarray                     # (n,)
    v: np.ndarray                     # (n,)
    this_mu: np.ndarray               # (n,)
    this_M: np.ndarray                # (n, k) Fortran-contiguous like the reference's
    this_omega2: np.ndarray           # (n,)
    unmasked_wavelengths: np.ndarray  # (n_u,)
    padded_wavelengths: np.ndarray    # (n_u + 6,)
    keep: np.ndarray                  # (n_u,) bool; keep.sum() == n
    injected_z: Tuple[float, ...]
    injected_log_nhi: Tuple[float, ...]

    @property
    def this_wavelengths(self) -> np.ndarray:
        return self.unmasked_wavelengths[self.keep]

    @property
    def n(self) -> int:
        return int(self.y.shape[0])

    @property
    def k(self) -> int:
        return int(self.this_M.shape[1])

    def z_range(self) -> Tuple[float, float]:
        """absorber redshift range searched (stand-in for set_parameters.py:125-159)"""
        lam = self.unmasked_wavelengths
        z_min = lam[0] / LYA_ANGSTROM - 1 + 0.01
        z_max = min(lam[-1] / LYA_ANGSTROM - 1, self.z_qso) - 0.01
        return z_min, z_max


def pad_wavelengths(unmasked: np.ndarray, width: int = 3, pixel_spacing: float = 1e-4) -> np.ndarray:
    """`width` log-spaced pixels on either side (reference null_gp.py:159-177)."""
    lo = np.log10(unmasked.min())
    hi = np.log10(unmasked.max())
    return np.concatenate(
        [
            np.logspace(lo - width * pixel_spacing, lo - pixel_spacing, width),
            unmasked,
            np.logspace(hi + pixel_spacing, hi + width * pixel_spacing, width),
        ]
    )


def _host_absorption(padded: np.ndarray, z_dla: float, nhi: float, num_lines: int) -> np.ndarray:
    """Voigt absorption + instrument convolution, used only to inject absorbers into y."""
    scale = np.sqrt(2) * ls.SIGMA_CGS
    tau = np.zeros_like(padded)
    for line in range(num_lines):
        velocity = padded * (ls.SPEED_OF_LIGHT_CGS / (ls.TRANSITION_WAVELENGTHS[line] * (1 + z_dla)) / 1e8)
        velocity = velocity - ls.SPEED_OF_LIGHT_CGS
        profile = np.real(wofz((velocity + 1j * ls.LORENTZ_WIDTHS[line]) / scale))
        tau += ls.LEADING_CONSTANTS[line] * profile / (np.sqrt(2 * np.pi) * ls.SIGMA_CGS)
    return np.convolve(np.exp(-nhi * tau), ls.INSTRUMENT_PROFILE, "valid")


def make_sightline(
    index: int = 0,
    n_unmasked: int = 3061,
    n_masked: int = 61,
    k: int = 20,
    z_qso: float = 3.0,
    num_lines: int = 3,
    injected: Optional[Sequence[Tuple[float, float]]] = None,
) -> SyntheticSightline:
    """
    Sightline `index` of the synthetic workload.  Defaults give BASELINE config 2:
    n_u = 3061 unmasked pixels, 61 masked -> n = 3000 kept pixels, rank k = 20.

    `injected` = [(z_dla, log_nhi), ...]; default injects (index mod 3) absorbers at
    fixed positions so that model selection has a non-trivial answer.
    """
    rng = np.random.default_rng(BASE_SEED + index)
    j = np.arange(n_unmasked, dtype=np.float64)
    unmasked = 3647.0 * 10.0 ** (1e-4 * j)
    padded = pad_wavelengths(unmasked)

    keep = np.ones(n_unmasked, dtype=bool)
    if n_masked:
        keep[rng.choice(n_unmasked, size=n_masked, replace=False)] = False
    jj = j[keep]
    n = jj.shape[0]

    scale = n_unmasked / 3061.0
    mu = 1 + 0.3 * np.sin(2 * np.pi * jj / (700 * scale)) + 0.5 * np.exp(-(((jj - 2600 * scale) / (60 * scale)) ** 2))
    box = np.ones(min(25, n)) / min(25, n)  # 25-pixel boxcar (shorter only for toy sightlines of fewer pixels)
    M = np.empty((k, n))
    for col in range(k):
        M[col] = 0.15 * np.convolve(rng.standard_normal(n), box, "same") * 0.9 ** col
    M = M.T  # (n, k), Fortran-contiguous as in the reference (null_gp.py:88-93)
    omega2 = 0.01 * (1 + rng.random(n))
    v = 0.02 * (1 + 3 * rng.random(n))

    if injected is None:
        lam0, lam1 = unmasked[0], unmasked[-1]
        z_lo = lam0 / LYA_ANGSTROM - 1 + 0.01
        z_hi = min(lam1 / LYA_ANGSTROM - 1, z_qso) - 0.01
        candidates = [(z_lo + 0.37 * (z_hi - z_lo), 20.8), (z_lo + 0.71 * (z_hi - z_lo), 21.3)]
        injected = candidates[: index % 3]
    a_true = np.ones(n_unmasked)
    for z_dla, log_nhi in injected:
        a_true = a_true * _host_absorption(padded, z_dla, 10.0 ** log_nhi, num_lines)
    a_true = a_true[keep]

    y = mu * a_true + M @ rng.standard_normal(k) + np.sqrt(omega2 * a_true ** 2 + v) * rng.standard_normal(n)

    return SyntheticSightline(
        z_qso=z_qso,
        y=y,
        v=v,
        this_mu=mu,
        this_M=M,
        this_omega2=omega2,
        unmasked_wavelengths=unmasked,
        padded_wavelengths=padded,
        keep=keep,
        injected_z=tuple(z for z, _ in injected),
        injected_log_nhi=tuple(l for _, l in injected),
    )


def make_qmc_samples(sight: SyntheticSightline, num: int = 10000, sub_dla: bool = False):
    """
    (z_dla (num,), log_nhi (num,), nhi (num,)): Halton base 2 / base 3.
    DLA: log N = 20 + 3 u; sub-DLA: log N = 19.5 + 0.5 u.
    """
    z_min, z_max = sight.z_range()
    z = z_min + (z_max - z_min) * halton(num, 2)
    u = halton(num, 3)
    log_nhi = (19.5 + 0.5 * u) if sub_dla else (20.0 + 3.0 * u)
    return z, log_nhi, 10.0 ** log_nhi


def make_catalogue(num, k=20, seed=7, mix="uniform"):
    """
    A synthetic catalogue of RAW spectra for the catalogue driver (`run_bayes_select.process_qso`): a learned model on the
    reference's rest-frame grid (set_parameters.py:33-35), SDSS-like spectra (4600 pixels at 1e-4 dex from 3631 A, 3 % of
    them masked, an absorber in two of three) and their quasar redshifts.  The kept pixels n of a spectrum follow z_qso
    through the modelled window 911.75-1215.75 A (null_gp.py:139-152).  `mix`: "uniform" z_qso in [2.2, 3.4] (n = 300 ... 1250,
    mean ~ 920) or "dr12q" z_qso = 2.15 + Gamma(2, 0.22) (n = 250 ... 1250, median ~ 685).
    :return: (LearnedModel, [(wavelengths, flux, noise_variance, pixel_mask)], [z_qso])
    """
    from .run_bayes_select import LearnedModel

    rng = np.random.default_rng(seed)
    rest_grid = np.arange(911.75, 1215.75 + 1e-9, 0.25)
    mu = 1 + 0.2 * np.cos(rest_grid / 30.0)
    M = 0.1 * rng.standard_normal((rest_grid.shape[0], k)) * (0.9 ** np.arange(k))[None, :]
    log_omega = -2.0 + 0.3 * np.sin(rest_grid / 55.0)
    learned = LearnedModel(rest_grid, mu, M, log_omega, -2.3, -5.9, 1.2)
    obs = 10.0 ** (3.5600 + 1e-4 * np.arange(4600))
    spectra, z_qsos = [], []
    for i in range(num):
        if mix == "dr12q":
            z_qso = float(np.clip(2.15 + rng.gamma(2.0, 0.22), 2.15, 5.5))
        else:
            z_qso = 2.2 + 1.2 * rng.random()
        rest = obs / (1 + z_qso)
        flux = 3.0 + 0.5 * np.sin(rest / 40.0)
        if i % 3:
            centre = 1215.6701 * (1 + z_qso - 0.3)
            flux = flux * np.exp(-50.0 / (1.0 + ((obs - centre) / 3.0) ** 2))
        flux = flux + 0.2 * rng.standard_normal(obs.shape[0])
        spectra.append((obs, flux, 0.04 * (1 + rng.random(obs.shape[0])), rng.random(obs.shape[0]) < 0.03))
        z_qsos.append(z_qso)
    return learned, spectra, z_qsos
