"""
Null (no absorber) GP model of the quasar emission, evaluated on the GPU.

Same constructor, `set_data` and attribute surface as the reference class
(gpy_dla_detection/null_gp.py:15-387) so that `BayesModelSelect` and
`run_bayes_select.process_qso` can use it unchanged.  The data/model preparation
(`set_data`, `get_interp`) is host NumPy -- it runs once per spectrum; the model
evidence itself goes through libdlagp_hip.so (`log_model_evidence` ->
dlagp_null, `log_mvnpdf_low_rank` -> dlagp_mvnpdf).
"""
from typing import Optional

import numpy as np
from scipy import interpolate

from .effective_optical_depth import effective_optical_depth
from .set_parameters import Parameters
from . import sightline as _sl


def prepare_data(params: Parameters, X: np.ndarray, Y: np.ndarray, noise_variance: np.ndarray,
                 pixel_mask: np.ndarray, z_qso: float, normalize: bool = True) -> dict:
    """
    The data half of `NullGP.set_data` (reference null_gp.py:117-177), model-free: normalisation,
    wavelength cuts, masks and the padded grid of the instrument convolution.  Returns the
    attributes `set_data` leaves on the instance, by name.
    """
    p = params
    if p.width != 3:
        # voigt.py:211-224: `width` = 3 is the half-width of the 7-tap SDSS instrument profile the device kernels
        # are written for (DLAGP_NUM_TAPS); the reference's own profile table has no other width either
        raise ValueError("Parameters.width = {} is not supported: the instrument profile has 2 * 3 + 1 taps".format(p.width))
    x, y, v = X, Y, noise_variance
    out = {"pixel_mask": pixel_mask}
    if normalize:
        window = (x >= p.normalization_min_lambda) & (x <= p.normalization_max_lambda) & (~pixel_mask)
        median = np.nanmedian(y[window])
        y = y / median
        v = v / median ** 2
        out["normalization_median"] = median

    in_range = (x >= p.min_lambda) & (x <= p.max_lambda)
    observed = Parameters.observed_wavelengths(x, z_qso)
    # complete, equally spaced grid for the absorption profile
    unmasked_wavelengths = observed[in_range]
    kept = in_range & (~pixel_mask)
    out.update(
        ind_unmasked=in_range, unmasked_wavelengths=unmasked_wavelengths, this_wavelengths=observed[kept],
        x=x[kept], y=y[kept], v=v[kept], ind=kept,
    )
    # `width` extra log-spaced pixels on either side for the instrument convolution
    lo = np.log10(unmasked_wavelengths.min())
    hi = np.log10(np.max(unmasked_wavelengths))
    out["padded_wavelengths"] = np.concatenate(
        [
            np.logspace(lo - p.width * p.pixel_spacing, lo - p.pixel_spacing, p.width),
            unmasked_wavelengths,
            np.logspace(hi + p.pixel_spacing, hi + p.width * p.pixel_spacing, p.width),
        ]
    )
    return out


class _LinearInterpolant:
    """
    `scipy.interpolate.interp1d(grid, values)` (kind "linear", bounds_error) as the reference builds it
    (null_gp.py:67-71), without the per-object set-up: the reference's driver constructs three GP objects per
    spectrum, 22 interpolants each, which was a quarter of the per-spectrum time of the drop-in route.  Same
    arithmetic -- interp1d's linear path is `np.interp` on a sorted grid -- and the same ValueError outside the grid.
    """

    __slots__ = ("x", "y")

    def __init__(self, grid: np.ndarray, values: np.ndarray):
        self.x = np.asarray(grid, dtype=np.float64)
        self.y = np.asarray(values, dtype=np.float64)
        if self.x.ndim != 1 or self.x.shape != self.y.shape:
            raise ValueError("x and y arrays must be equal in length along interpolation axis.")

    def __call__(self, x_new) -> np.ndarray:
        x_new = np.asarray(x_new, dtype=np.float64)
        if x_new.size:
            if np.min(x_new) < self.x[0]:
                raise ValueError("A value in x_new is below the interpolation range's minimum value ({}).".format(self.x[0]))
            if np.max(x_new) > self.x[-1]:
                raise ValueError("A value in x_new is above the interpolation range's maximum value ({}).".format(self.x[-1]))
        return np.interp(x_new, self.x, self.y)


class _InterpCache:
    """
    The reference's driver builds three GP objects per spectrum (null, sub-DLA, DLA model; run_bayes_select.py:151-186)
    and calls `set_data` on each, so `get_interp` interpolates the same learned model onto the same pixels three times
    (22 interpolations and two effective-optical-depth tables: 0.5 ms, a third of the drop-in route's time per
    spectrum).  The last few results are kept, keyed by the identity of the learned arrays (held here, so an id cannot
    be reused), their scalars, and the CONTENT of the spectrum's pixels; every caller gets its own copies.
    """

    def __init__(self, size: int = 4):
        import threading

        self.size = size
        self.entries = []  # [(key, held arrays, (this_mu, this_M, this_omega2))], most recent last
        self.lock = threading.Lock()

    def get(self, key):
        with self.lock:
            for i, (k, _, value) in enumerate(self.entries):
                if k == key:
                    self.entries.append(self.entries.pop(i))
                    return tuple(a.copy(order="K") for a in value)
        return None

    def put(self, key, held, value):
        with self.lock:
            self.entries.append((key, held, tuple(a.copy(order="K") for a in value)))
            del self.entries[: -self.size]


_INTERP_CACHE = _InterpCache()


class NullGP:
    """
    p(y | lambda, z_qso, v, omega, M) = N(y; mu a_lya, A_lya (M M' + Omega) A_lya + V)

    :param params: pipeline parameters
    :param prior: prior catalogue (only `less_ind(z_qso)` is used, by `log_prior`)
    :param rest_wavelengths: rest-frame grid the model was learned on
    :param mu: mean of the GP on that grid
    :param M: (len(grid), k) low-rank factor of the covariance
    :param log_omega: log of the pixel-wise absorption noise
    :param log_c_0, log_tau_0, log_beta: Lyman-series noise model parameters
    :param prev_tau_0, prev_beta: effective optical depth used for mean-flux suppression
    :param device: HIP device ordinal the evidence is evaluated on (extension; default 0)
    """

    def __init__(
        self,
        params: Parameters,
        prior,
        rest_wavelengths: np.ndarray,
        mu: np.ndarray,
        M: np.ndarray,
        log_omega: np.ndarray,
        log_c_0: float,
        log_tau_0: float,
        log_beta: float,
        prev_tau_0: float = 0.0023,
        prev_beta: float = 3.65,
        device: int = 0,
    ):
        self.params = params
        self.prior = prior

        self.rest_wavelengths = rest_wavelengths
        self.mu = mu
        self.M = M
        self.log_omega = log_omega
        self.log_c_0 = log_c_0
        self.log_tau_0 = log_tau_0
        self.log_beta = log_beta

        self.prev_tau_0 = prev_tau_0
        self.prev_beta = prev_beta

        self.device = device
        self._device_sightline: Optional[_sl.DeviceSightline] = None
        self._device_key = None

        # linear interpolants of the learned model (reference :67-71; scipy's interp1d when the grid is unsorted)
        if np.all(np.diff(np.asarray(rest_wavelengths, dtype=np.float64)) > 0):
            make = _LinearInterpolant
        else:
            make = interpolate.interp1d
        self.mu_interpolator = make(rest_wavelengths, mu)
        self.log_omega_interpolator = make(rest_wavelengths, log_omega)
        self.list_M_interpolators = [make(rest_wavelengths, column) for column in np.asarray(M).T]

    # ------------------------------------------------------------------------------------------
    # data / model preparation (host; reference :73-242)
    # ------------------------------------------------------------------------------------------
    def M_interpolator(self, this_rest_wavelengths: np.ndarray) -> np.ndarray:
        """(n, k) low-rank factor interpolated column by column; Fortran-ordered like the reference's"""
        assert len(self.list_M_interpolators) == self.params.k
        rows = np.empty((self.params.k, this_rest_wavelengths.shape[0]))
        for c, column in enumerate(self.list_M_interpolators):
            rows[c, :] = column(this_rest_wavelengths)
        return rows.T

    def set_data(
        self,
        X: np.ndarray,
        Y: np.ndarray,
        noise_variance: np.ndarray,
        pixel_mask: np.ndarray,
        z_qso: float,
        normalize: bool = True,
        build_model: bool = True,
    ) -> None:
        """
        Load one spectrum (reference :95-177).

        :param X: rest-frame wavelengths of the pixels
        :param Y: flux
        :param noise_variance: instrumental noise variance per pixel
        :param pixel_mask: True for pixels to ignore
        :param z_qso: quasar redshift
        """
        self._invalidate_device()
        self.z_qso = z_qso
        state = prepare_data(self.params, X, Y, noise_variance, pixel_mask, z_qso, normalize)
        for name, value in state.items():
            setattr(self, name, value)
        if build_model:
            self.get_interp(self.x, self.y, self.this_wavelengths, self.z_qso)

    def get_interp(self, x: np.ndarray, y: np.ndarray, wavelengths: np.ndarray, z_qso: float) -> None:
        """
        Interpolate the learned model onto the kept pixels and apply the Lyman-series
        mean-flux suppression and noise scaling (reference :179-242).
        """
        self._invalidate_device()
        learned = (self.rest_wavelengths, self.mu, self.M, self.log_omega)
        x = np.asarray(x)
        wavelengths = np.asarray(wavelengths)
        key = None
        if type(self).M_interpolator is NullGP.M_interpolator and all(isinstance(a, np.ndarray) for a in learned):
            key = (
                tuple(id(a) for a in learned), tuple(a.shape for a in learned),
                float(self.mu.sum()), float(self.M.sum()), float(self.log_omega.sum()),  # edits in place
                float(self.log_c_0), float(self.log_tau_0), float(self.log_beta), float(self.prev_tau_0),
                float(self.prev_beta), int(self.params.num_forest_lines), int(self.params.k), float(z_qso),
                x.dtype.str, x.tobytes(), wavelengths.dtype.str, wavelengths.tobytes(),
            )
            hit = _INTERP_CACHE.get(key)
            if hit is not None:
                self.this_mu, self.this_M, self.this_omega2 = hit
                return
        this_mu = self.mu_interpolator(x)
        this_M = self.M_interpolator(x)
        this_omega2 = np.exp(2 * self.log_omega_interpolator(x))

        # mean-flux suppression from the Lyman-series forest
        suppression_tau = effective_optical_depth(
            wavelengths, self.prev_beta, self.prev_tau_0, z_qso, self.params.num_forest_lines
        )
        lya_absorption = np.exp(-np.sum(suppression_tau, axis=1))
        this_mu = this_mu * lya_absorption
        this_M = this_M * lya_absorption[:, None]
        assert this_M.shape[1] == self.params.k

        # redshift-dependent absorption noise
        noise_tau = effective_optical_depth(
            wavelengths, np.exp(self.log_beta), np.exp(self.log_tau_0), z_qso, self.params.num_forest_lines
        )
        scaling = 1 - np.exp(-np.sum(noise_tau, axis=1)) + np.exp(self.log_c_0)
        this_omega2 = this_omega2 * scaling ** 2
        this_omega2 = this_omega2 * lya_absorption ** 2

        self.this_mu = this_mu
        self.this_M = this_M
        self.this_omega2 = this_omega2
        if key is not None:
            _INTERP_CACHE.put(key, learned, (this_mu, this_M, this_omega2))

    # ------------------------------------------------------------------------------------------
    # views (reference :244-292)
    # ------------------------------------------------------------------------------------------
    @property
    def mean(self):
        return self.mu

    @property
    def K(self):
        return np.matmul(self.M, self.M.T)

    @property
    def this_mean(self):
        return self.this_mu

    @property
    def this_noise(self):
        return self.this_omega2 + self.v

    @property
    def this_K(self):
        return np.matmul(self.this_M, self.this_M.T)

    @property
    def X(self) -> np.ndarray:
        return self.x

    @property
    def Y(self) -> np.ndarray:
        return self.y

    @property
    def V(self) -> np.ndarray:
        return self.v

    # ------------------------------------------------------------------------------------------
    # device residency
    # ------------------------------------------------------------------------------------------
    def _invalidate_device(self) -> None:
        if self._device_sightline is not None:
            self._device_sightline.close()
        self._device_sightline = None
        self._device_key = None

    def _keep_flags(self) -> np.ndarray:
        return ~self.pixel_mask[self.ind_unmasked]  # reference dla_gp.py:360

    def device_sightline(self) -> _sl.DeviceSightline:
        """
        The HBM-resident copy of this instance's post-set_data state.  Rebuilt when the arrays the
        sweep reads were replaced (set_data / get_interp / attribute assignment) or when
        `params.num_lines` / `broadening` changed.
        """
        broadening = bool(getattr(self, "broadening", True))
        # identity (not equality) of the arrays: in-place edits of a loaded array are not tracked
        arrays = (
            self.y, self.v, self.this_mu, self.this_M, self.this_omega2, self.padded_wavelengths,
            self.unmasked_wavelengths, self.pixel_mask, self.ind_unmasked,
        )
        settings = (int(self.params.num_lines), broadening, int(self.device), bool(getattr(self, "lls_break", False)))
        stale = (
            self._device_sightline is None
            or self._device_key[1] != settings
            or any(a is not b for a, b in zip(self._device_key[0], arrays))
        )
        if stale:
            self._invalidate_device()
            grid = self.padded_wavelengths if broadening else self.unmasked_wavelengths
            self._device_sightline = _sl.DeviceSightline(
                self.y, self.this_mu, self.this_M, self.this_omega2, self.v, grid, self._keep_flags(),
                num_lines=settings[0], broadening=broadening, device=settings[2], lls_break=settings[3],
            )
            self._device_key = (arrays, settings)
        return self._device_sightline

    # ------------------------------------------------------------------------------------------
    # model evidence (reference :294-360) -- GPU
    # ------------------------------------------------------------------------------------------
    def log_model_evidence(self) -> float:
        """log p(y | no absorber) for the loaded spectrum"""
        return self.device_sightline().null()

    @staticmethod
    def log_mvnpdf_low_rank(y: np.ndarray, mu: np.ndarray, M: np.ndarray, d: np.ndarray,
                            scipy_lapack: bool = True) -> float:
        """
        log N(y; mu, M M' + diag(d)) through the Woodbury identity, on the GPU.

        `scipy_lapack` is accepted for signature compatibility; both reference branches
        compute the same quantity and there is a single device implementation.  Ranks up to
        DLAGP_MAX_RANK = 64 (include/dlagp.h; the reference's learned models have k = 20); a larger k raises
        ValueError.
        """
        return _sl.log_mvnpdf_low_rank(y, mu, M, d)

    def log_prior(self, z_qso: float, without_subDLAs: bool = True) -> float:
        """log P(no DLA | z_qso) = log(1 - M/N) (reference :362-387)"""
        num_dlas, num_quasars = self.prior.less_ind(z_qso)
        return np.log(1 - (num_dlas / num_quasars))

    # ------------------------------------------------------------------------------------------
    @classmethod
    def from_mat(cls, params: Parameters, prior,
                 learned_file: str = "learned_qso_model_lyseries_variance_kim_dr9q_minus_concordance.mat",
                 **kwargs):
        """Load the learned model from the MATLAB v7.3 file (reference NullGPMAT, :390-422)."""
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
        return cls(params, prior, *args, prev_tau_0=0.0023, prev_beta=3.65, **kwargs)


def NullGPMAT(params: Parameters, prior,
              learned_file: str = "learned_qso_model_lyseries_variance_kim_dr9q_minus_concordance.mat"):
    return NullGP.from_mat(params, prior, learned_file)
