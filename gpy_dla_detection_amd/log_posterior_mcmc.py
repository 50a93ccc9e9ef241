"""
Log posterior of absorber parameters for MCMC refinement, evaluated on the GPU.

Mirror of the reference's functional interface (gpy_dla_detection/log_posterior_mcmc.py:17-250):
`log_prior`, `log_posterior(theta, ...)` with the argument list `DLAGP.run_mcmc` hands to
emcee (dla_gp.py:265-287), and `sample_log_likelihood_k_dlas`.  The likelihood is the same
device path as the QMC sweep (`dlagp_eval`); the sightline context is built from the arrays on
first use and reused while arrays of the same CONTENT are passed (emcee calls with identical `args`).
The reference re-reads its arrays on every call (:139-250), so an in-place edit of `y` between two
calls changes its answer; the cache key is therefore the arrays' bytes (a memcmp of a few hundred KB
per call), not their identity, and the cache is guarded by a lock (emcee pools may call from threads).

`log_posterior_vectorized` is the form for `emcee.EnsembleSampler(..., vectorize=True)`: all
walkers of an ensemble step in one launch -- the way to use the GPU for MCMC.  Unlike the
reference's TODO (:66), theta may hold k absorbers: (z_1..z_k, logN_1..logN_k).
"""
from typing import Callable, Optional, Tuple

import threading

import numpy as np

from . import sightline as _sl
from .null_gp import _content

_cache = {"key": None, "dev": None}
_cache_lock = threading.Lock()  # the cached sightline is built, replaced and evaluated under it (a context is not thread-safe)


def log_prior(z_dla: float, log_nhi: float, min_z_dla: float, max_z_dla: float, min_log_nhi: float,
              max_log_nhi: float, pdf: Callable) -> float:
    """uniform in z_dla inside the search range, `pdf` in log N_HI (reference :17-43)"""
    inside = (z_dla < max_z_dla) * (z_dla > min_z_dla) * (log_nhi > min_log_nhi) * (log_nhi < max_log_nhi)
    if inside:
        return np.log(pdf(log_nhi))
    return -np.inf


def _device_sightline(y, v, padded_wavelengths, this_mu, this_M, this_omega2, pixel_mask, ind_unmasked,
                      num_lines, device: int = 0) -> _sl.DeviceSightline:
    """the cached HBM copy of these arrays, rebuilt whenever their CONTENT differs; call with `_cache_lock` held"""
    key = (int(num_lines), int(device),
           _content(y, v, padded_wavelengths, this_mu, this_M, this_omega2, pixel_mask, ind_unmasked))
    if _cache["dev"] is None or _cache["key"] != key:
        if _cache["dev"] is not None:
            _cache["dev"].close()
            _cache["dev"] = _cache["key"] = None
        keep = ~np.asarray(pixel_mask)[np.asarray(ind_unmasked)]  # reference :196
        _cache["dev"] = _sl.DeviceSightline(y, this_mu, this_M, this_omega2, v, padded_wavelengths, keep,
                                            num_lines=int(num_lines), broadening=True, device=device)
        _cache["key"] = key
    return _cache["dev"]


def _evaluate(z: np.ndarray, nhi: np.ndarray, arrays, num_lines, device: int = 0) -> np.ndarray:
    """log-likelihoods of the rows of (z, nhi) on the cached sightline of `arrays`"""
    with _cache_lock:
        return _device_sightline(*arrays, num_lines, device).eval(z, nhi)


def sample_log_likelihood_k_dlas(z_dlas: np.ndarray, nhis: np.ndarray, y: np.ndarray, v: np.ndarray,
                                 padded_wavelengths: np.ndarray, this_mu: np.ndarray, this_M: np.ndarray,
                                 this_omega2: np.ndarray, pixel_mask: np.ndarray, ind_unmasked: np.ndarray,
                                 num_lines: int) -> float:
    """log p(y | {z_dla, N_HI}_1..k) (reference :98-136); the broadened profile, as there (:179-186)"""
    assert len(z_dlas) == len(nhis)
    z = np.asarray(z_dlas, dtype=np.float64)[None, :]
    n = np.asarray(nhis, dtype=np.float64)[None, :]
    arrays = (y, v, padded_wavelengths, this_mu, this_M, this_omega2, pixel_mask, ind_unmasked)
    return float(_evaluate(z, n, arrays, num_lines)[0])


def log_posterior(theta: Tuple, this_wavelengths: np.ndarray, y: np.ndarray, v: np.ndarray, z_qso: float,
                  min_z_dla: float, max_z_dla: float, min_log_nhi: float, max_log_nhi: float, pdf: Callable,
                  padded_wavelengths: np.ndarray, this_mu: np.ndarray, this_M: np.ndarray, this_omega2: np.ndarray,
                  pixel_mask: np.ndarray, ind_unmasked: np.ndarray, num_lines: int) -> float:
    """log p(theta | y) + const for one parameter vector (reference :46-95)"""
    out = log_posterior_vectorized(
        np.asarray(theta, dtype=np.float64)[None, :], this_wavelengths, y, v, z_qso, min_z_dla, max_z_dla,
        min_log_nhi, max_log_nhi, pdf, padded_wavelengths, this_mu, this_M, this_omega2, pixel_mask,
        ind_unmasked, num_lines)
    return float(out[0])


def log_posterior_vectorized(thetas: np.ndarray, this_wavelengths: np.ndarray, y: np.ndarray, v: np.ndarray,
                             z_qso: float, min_z_dla: float, max_z_dla: float, min_log_nhi: float,
                             max_log_nhi: float, pdf: Callable, padded_wavelengths: np.ndarray,
                             this_mu: np.ndarray, this_M: np.ndarray, this_omega2: np.ndarray,
                             pixel_mask: np.ndarray, ind_unmasked: np.ndarray, num_lines: int,
                             device: Optional[int] = 0) -> np.ndarray:
    """
    `log_posterior` of every row of `thetas` (W, 2k) = (z_1..z_k, logN_1..logN_k): priors on the host,
    the likelihoods of all rows with a finite prior in one device launch.
    """
    thetas = np.atleast_2d(np.asarray(thetas, dtype=np.float64))
    W, width = thetas.shape
    assert width % 2 == 0
    k = width // 2
    z, log_nhi = thetas[:, :k], thetas[:, k:]
    lp = np.zeros(W)
    for w in range(W):
        for j in range(k):
            lp[w] += log_prior(z[w, j], log_nhi[w, j], min_z_dla, max_z_dla, min_log_nhi, max_log_nhi, pdf)
    out = np.full(W, -np.inf)
    live = np.isfinite(lp)
    if live.any():
        arrays = (y, v, padded_wavelengths, this_mu, this_M, this_omega2, pixel_mask, ind_unmasked)
        out[live] = lp[live] + _evaluate(np.ascontiguousarray(z[live]), np.ascontiguousarray(10.0 ** log_nhi[live]),
                                         arrays, num_lines, device)
    return out
