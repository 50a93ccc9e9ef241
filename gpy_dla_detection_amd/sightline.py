"""
Python handles on the HBM-resident sightline contexts of libdlagp_hip.so.

`DeviceSightline` owns one `dlagp_ctx` (the post-`set_data` state of one spectrum,
reference null_gp.py:117-177, :240-242); `DeviceBatch` groups several for one launch.
The GP classes (null_gp.py, dla_gp.py, subdla_gp.py) and bench.py call through these.
"""
import ctypes
from typing import Optional, Sequence

import numpy as np

from . import _lib


class DeviceModel:
    """
    The learned GP model on its rest-frame grid, resident on one device and shared by every
    spectrum interpolated from it (arguments of NullGP.__init__, reference null_gp.py:35-71).
    """

    def __init__(self, rest_wavelengths, mu, M, log_omega, log_c_0: float, log_tau_0: float, log_beta: float,
                 prev_tau_0: float = 0.0023, prev_beta: float = 3.65, num_forest_lines: int = 31, device: int = 0):
        lib = _lib.load()
        grid, mu, log_omega = (_lib.as_f64(a) for a in (rest_wavelengths, mu, log_omega))
        M = _lib.strided_f64(M)
        if M.ndim != 2 or M.shape[0] != grid.shape[0] or mu.shape != grid.shape or log_omega.shape != grid.shape:
            raise ValueError("model arrays must be (n_grid,), (n_grid,), (n_grid, k), (n_grid,)")
        handle = ctypes.c_void_p()
        _lib.check(lib.dlagp_model_create(
            ctypes.byref(handle), grid.shape[0], M.shape[1], _lib.ptr(grid), _lib.ptr(mu), _lib.ptr(M),
            M.strides[0] // 8, M.strides[1] // 8, _lib.ptr(log_omega), float(log_c_0), float(log_tau_0),
            float(log_beta), float(prev_tau_0), float(prev_beta), int(num_forest_lines), int(device)))
        self._handle = handle
        self.k, self.device = M.shape[1], int(device)

    def close(self) -> None:
        if getattr(self, "_handle", None):
            _lib.load().dlagp_model_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceSightline:
    """
    :param y, mu, omega2, v: (n,) flux, GP mean, absorption noise, instrumental noise variance
    :param M: (n, k) low-rank factor, any element strides
    :param wavelengths: (n_u + 6,) padded observed wavelengths if `broadening`, else (n_u,)
    :param keep: (n_u,) bool, True for the n kept pixels (= ~pixel_mask[ind_unmasked])
    """

    def __init__(self, y, mu, M, omega2, v, wavelengths, keep, num_lines: int = 3,
                 broadening: bool = True, device: int = 0, lls_break: bool = False):
        lib = _lib.load()
        y, mu, omega2, v = (_lib.as_f64(a) for a in (y, mu, omega2, v))
        wavelengths = _lib.as_f64(wavelengths)
        M = _lib.strided_f64(M)
        keep = np.ascontiguousarray(keep, dtype=np.uint8)
        if M.ndim != 2:
            raise ValueError("M must be (n, k)")
        n, k = M.shape
        n_u = keep.shape[0]
        for name, arr in (("y", y), ("mu", mu), ("omega2", omega2), ("v", v)):
            if arr.shape != (n,):
                raise ValueError("{} has shape {}, expected ({},)".format(name, arr.shape, n))
        expected = n_u + 6 if broadening else n_u
        if wavelengths.shape != (expected,):
            raise ValueError("wavelengths has shape {}, expected ({},)".format(wavelengths.shape, expected))
        handle = ctypes.c_void_p()
        rc = lib.dlagp_create(
            ctypes.byref(handle), n, n_u, k, _lib.ptr(y), _lib.ptr(mu), _lib.ptr(M),
            M.strides[0] // 8, M.strides[1] // 8, _lib.ptr(omega2), _lib.ptr(v), _lib.ptr(wavelengths),
            keep.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), int(num_lines),
            int(bool(broadening)) | (2 if lls_break else 0), int(device),
        )
        _lib.check(rc)
        self._handle = handle
        self.n, self.k, self.device = n, k, int(device)
        self.num_lines, self.broadening, self.lls_break = int(num_lines), bool(broadening), bool(lls_break)

    @classmethod
    def from_model(cls, model: DeviceModel, x, y, v, this_wavelengths, z_qso: float, wavelengths, keep,
                   num_lines: int = 3, broadening: bool = True, lls_break: bool = False, defer_sync: bool = False):
        """
        NullGP.get_interp on the device (dlagp_create_from_model): the GP mean, factor and absorption
        noise are interpolated from `model` at the kept pixels' rest wavelengths `x` and suppressed by
        the Lyman-series mean flux at the observed `this_wavelengths` (reference null_gp.py:179-242).

        `defer_sync` (batch drivers): do not wait for the context's set-up work -- its first use (a `DeviceBatch`
        built with it, an evaluation, on any thread) waits for it: one wait per batch instead of one per spectrum.
        """
        lib = _lib.load()
        x, y, v, this_wavelengths = (_lib.as_f64(a) for a in (x, y, v, this_wavelengths))
        wavelengths = _lib.as_f64(wavelengths)
        keep = np.ascontiguousarray(keep, dtype=np.uint8)
        n, n_u = x.shape[0], keep.shape[0]
        for name, arr in (("y", y), ("v", v), ("this_wavelengths", this_wavelengths)):
            if arr.shape != (n,):
                raise ValueError("{} has shape {}, expected ({},)".format(name, arr.shape, n))
        expected = n_u + 6 if broadening else n_u
        if wavelengths.shape != (expected,):
            raise ValueError("wavelengths has shape {}, expected ({},)".format(wavelengths.shape, expected))
        handle = ctypes.c_void_p()
        _lib.check(lib.dlagp_create_from_model(
            ctypes.byref(handle), model._handle, n, n_u, _lib.ptr(x), _lib.ptr(y), _lib.ptr(v),
            _lib.ptr(this_wavelengths), float(z_qso), _lib.ptr(wavelengths),
            keep.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), int(num_lines),
            int(bool(broadening)) | (2 if lls_break else 0) | (4 if defer_sync else 0)))
        self = cls.__new__(cls)
        self._handle = handle
        self.n, self.k, self.device = n, model.k, model.device
        self.num_lines, self.broadening, self.lls_break = int(num_lines), bool(broadening), bool(lls_break)
        return self

    def read_model(self):
        """(this_mu (n,), this_M (n, k), this_omega2 (n,)) as they sit on the device"""
        mu = np.empty(self.n)
        M = np.empty((self.n, self.k))
        omega2 = np.empty(self.n)
        _lib.check(_lib.load().dlagp_ctx_read_model(self._handle, _lib.ptr(mu), _lib.ptr(M), _lib.ptr(omega2)))
        return mu, M, omega2

    @classmethod
    def from_state(cls, st, num_lines: int = 3, broadening: bool = True, device: int = 0, lls_break: bool = False):
        """from any object with the reference's post-set_data attribute names"""
        keep = getattr(st, "keep", None)
        if keep is None:
            keep = ~st.pixel_mask[st.ind_unmasked]
        grid = st.padded_wavelengths if broadening else st.unmasked_wavelengths
        return cls(st.y, st.this_mu, st.this_M, st.this_omega2, st.v, grid, keep,
                   num_lines=num_lines, broadening=broadening, device=device, lls_break=lls_break)

    # -- lifetime -----------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_handle", None):
            _lib.load().dlagp_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- evaluation (host arrays) -------------------------------------------------------------
    @staticmethod
    def _samples(z, nhi):
        z = _lib.as_f64(z)
        nhi = _lib.as_f64(nhi)
        if z.shape != nhi.shape:
            raise ValueError("z and nhi must have the same shape")  # assert at dla_gp.py:321
        if z.ndim == 1:
            z, nhi = z[:, None], nhi[:, None]
        if z.ndim != 2:
            raise ValueError("samples must be (S,) or (S, kd)")
        return np.ascontiguousarray(z), np.ascontiguousarray(nhi)

    def eval(self, z, nhi) -> np.ndarray:
        """log-likelihood of S samples with kd absorbers each; z, nhi: (S,) or (S, kd)"""
        z, nhi = self._samples(z, nhi)
        S, kd = z.shape
        out = np.empty(S, dtype=np.float64)
        _lib.check(_lib.load().dlagp_eval(self._handle, S, kd, _lib.ptr(z), _lib.ptr(nhi), _lib.ptr(out)))
        return out

    def null(self) -> float:
        out = np.empty(1, dtype=np.float64)
        _lib.check(_lib.load().dlagp_null(self._handle, _lib.ptr(out)))
        return float(out[0])

    def absorption(self, z_dlas, nhis) -> np.ndarray:
        """absorption vector on the kept pixels for one sample (dla_gp.py:360-388)"""
        z = _lib.as_f64(np.atleast_1d(z_dlas))
        nhi = _lib.as_f64(np.atleast_1d(nhis))
        if z.shape != nhi.shape or z.ndim != 1:
            raise ValueError("z_dlas and nhis must be 1-d and of equal length")
        out = np.empty(self.n, dtype=np.float64)
        _lib.check(_lib.load().dlagp_absorption(self._handle, z.shape[0], _lib.ptr(z), _lib.ptr(nhi), _lib.ptr(out)))
        return out

    # -- evaluation (HBM-resident torch tensors or raw device pointers) -------------------------
    def eval_device(self, S: int, kd: int, d_z: int, d_nhi: int, d_out: int, stream: int = 0) -> None:
        """enqueue on `stream` (raw hipStream_t value); pointers are device addresses"""
        _lib.check(_lib.load().dlagp_eval_device(self._handle, S, kd, d_z, d_nhi, d_out, stream))

    def eval_evidence_device(self, S: int, kd: int, d_z: int, d_nhi: int, d_out: int, level: int, d_evidence: int,
                             d_too_close: int = 0, stream: int = 0) -> None:
        """
        `eval_device` and `level_evidence_device` in one call (dlagp_eval_evidence_device): d_out receives log p - log S
        (NaN where too_close), d_evidence the level's log-mean-exp evidence -- from the finalisation kernel's own tail where
        the launch has one, no second launch (reference dla_gp.py:132-190 for one level)
        """
        _lib.check(_lib.load().dlagp_eval_evidence_device(self._handle, S, kd, d_z, d_nhi, d_out, level, d_too_close or None,
                                                          d_evidence, stream))


def prepare_params(params, normalize: bool = True) -> "_lib.PrepareParams":
    """`Parameters` (reference set_parameters.py:14-165) as the native data half reads it (dlagp_prepare_params)"""
    return _lib.PrepareParams(
        float(params.min_lambda), float(params.max_lambda), float(params.normalization_min_lambda),
        float(params.normalization_max_lambda), float(params.pixel_spacing), float(params.lya_wavelength),
        float(params.lyman_limit), float(params.min_z_cut), float(params.max_z_cut), int(params.width), int(bool(normalize)))


def numpy_padding(width: int, pixel_spacing: float, edges: np.ndarray) -> np.ndarray:
    """
    The padding pixels of `nb` spectra, (nb, 2 width), from their smallest and largest unmasked wavelengths `edges` (nb, 2):
    `null_gp._logspace` (= np.logspace of reference null_gp.py:159-177) element by element -- np.log10, the same sums and
    products, np.power -- for the whole batch at once.  NumPy's own log10 / power on purpose: on AVX-512 hosts they are
    SVML's and differ from libm's in the last bit for a few per cent of the arguments.
    """
    out = np.empty((edges.shape[0], 2 * width))
    logs = np.log10(edges)
    for side, (start, stop) in enumerate(((logs[:, 0] - width * pixel_spacing, logs[:, 0] - pixel_spacing),
                                          (logs[:, 1] + pixel_spacing, logs[:, 1] + width * pixel_spacing))):
        step = (stop - start) / (width - 1)
        y = np.arange(0, width, dtype=np.float64)[None, :] * step[:, None]
        y += start[:, None]
        y[:, -1] = stop
        out[:, side * width:(side + 1) * width] = np.power(10.0, y)
    return out


class SpectrumTable:
    """
    A catalogue's raw spectra -- per quasar (observed wavelengths, flux, noise variance, pixel mask), what the reference's
    `read_spec` returns (run_bayes_select.py:147) -- as the native data half takes them: float64 / one-byte arrays and a
    table of their addresses, built once.  `ok` is False when an array is not float64 (the Python path serves: its
    arithmetic would run in that other type).
    """

    def __init__(self, spectra, z_qso_list):
        self.arrays = []  # keeps the (possibly converted) arrays alive
        num = len(spectra)
        self.addresses = np.zeros((4, num), dtype=np.uintp)
        self.num_pixels = np.zeros(num, dtype=np.int32)
        self.z_qso = np.ascontiguousarray(z_qso_list, dtype=np.float64)
        self.ok = True
        for q, (wavelengths, flux, noise_variance, pixel_mask) in enumerate(spectra):
            arrays = [np.asarray(a) for a in (wavelengths, flux, noise_variance)]
            mask = np.asarray(pixel_mask)
            if any(a.dtype != np.float64 or a.ndim != 1 for a in arrays) or mask.dtype != np.bool_ or mask.ndim != 1 \
                    or any(a.shape != mask.shape for a in arrays):
                self.ok = False
                return
            arrays = [np.ascontiguousarray(a) for a in arrays] + [np.ascontiguousarray(mask).view(np.uint8)]
            self.arrays.append(arrays)
            self.num_pixels[q] = mask.shape[0]
            for c, a in enumerate(arrays):
                self.addresses[c, q] = a.__array_interface__["data"][0]


def batch_prepare(model: "DeviceModel", table: SpectrumTable, rows, params_struct, num_lines: int, broadening: bool,
                  offsets: np.ndarray, sub_offsets: Optional[np.ndarray] = None, lls_break: bool = False):
    """
    The data half of `NullGP.set_data`, the contexts (`get_interp` on the device), the batch over them and the sample
    redshift tables of the spectra `rows` of `table`, in ONE call into the library with the interpreter lock released
    (dlagp_batch_prepare): what `null_gp.prepare_data`, `Parameters.z_dla_range`, `DeviceSightline.from_model` and
    `DeviceBatch` do per spectrum.

    :return: (DeviceBatch, [DeviceSightline], sample_z (nb, S), sub_sample_z (nb, S) or None when `sub_offsets` is None,
        scalars (nb, 8) = normalisation median, z_min, z_max (sample tables), z_min, z_max (raw spectrum), n, n_u, 0)
    """
    lib = _lib.load()
    rows = np.ascontiguousarray(rows, dtype=np.intp)
    nb = rows.shape[0]
    pointers = [np.ascontiguousarray(table.addresses[c, rows]) for c in range(4)]
    num_pixels = np.ascontiguousarray(table.num_pixels[rows])
    z_qso = np.ascontiguousarray(table.z_qso[rows])
    offsets = np.ascontiguousarray(_lib.as_f64(offsets))
    S = offsets.shape[0]
    sample_z = np.empty((nb, S))
    sub_sample_z = None
    if sub_offsets is not None:
        sub_offsets = np.ascontiguousarray(_lib.as_f64(sub_offsets))
        if sub_offsets.shape != (S,):
            raise ValueError("both sample holders need the same number of samples")
        sub_sample_z = np.empty((nb, S))
    scalars = np.empty((nb, 8))
    failure = []

    def pad(_user, count, width, spacing, edges, out):
        try:  # an exception cannot cross the C frames: it is kept and raised after the call
            np.ctypeslib.as_array(out, shape=(count, 2 * width))[:] = numpy_padding(
                width, spacing, np.ctypeslib.as_array(edges, shape=(count, 2)))
        except BaseException as exc:  # noqa: BLE001
            failure.append(exc)

    callback = _lib.PAD_CALLBACK(pad)
    batch_handle = ctypes.c_void_p()
    ctx_handles = (ctypes.c_void_p * nb)()
    as_pp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_void_p))  # noqa: E731
    rc = lib.dlagp_batch_prepare(
        ctypes.byref(batch_handle), ctx_handles, model._handle, ctypes.byref(params_struct), nb,
        num_pixels.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), as_pp(pointers[0]), as_pp(pointers[1]), as_pp(pointers[2]),
        as_pp(pointers[3]), _lib.ptr(z_qso), int(num_lines), int(bool(broadening)) | (2 if lls_break else 0), callback, None,
        S, _lib.ptr(offsets), _lib.ptr(sub_offsets) if sub_offsets is not None else None, _lib.ptr(sample_z),
        _lib.ptr(sub_sample_z) if sub_sample_z is not None else None, _lib.ptr(scalars))
    if failure and rc == _lib.DLAGP_OK:  # the padding never arrived: the contexts are built on garbage
        lib.dlagp_batch_destroy(batch_handle)
        for handle in ctx_handles:
            lib.dlagp_destroy(handle)
        raise failure[0]
    _lib.check(rc)
    sightlines = []
    for b in range(nb):
        sl = DeviceSightline.__new__(DeviceSightline)
        sl._handle = ctypes.c_void_p(ctx_handles[b])
        sl.n, sl.k, sl.device = int(scalars[b, 5]), model.k, model.device
        sl.num_lines, sl.broadening, sl.lls_break = int(num_lines), bool(broadening), bool(lls_break)
        sightlines.append(sl)
    batch = DeviceBatch.__new__(DeviceBatch)
    batch.sightlines, batch.nb, batch._handle = sightlines, nb, batch_handle
    batch._pending, batch.last_ties, batch.last_changed = None, None, 0
    return batch, sightlines, sample_z, sub_sample_z, scalars


def exact_choice(level_ll: np.ndarray, uniforms: np.ndarray) -> np.ndarray:
    """
    `np.random.choice(np.arange(S).astype(np.int32), size=S, replace=True, p=W / W.sum())` of the reference
    (dla_gp.py:209-218) for the uniforms its `random_sample(S)` would have drawn: W = exp(ll - nanmax), NaN -> 0, and the
    legacy generator's own steps -- `cdf = p.cumsum(); cdf /= cdf[-1]; cdf.searchsorted(uniforms, side="right")` -- in
    NumPy, so that every rounding is NumPy's.
    """
    weights = np.exp(level_ll - np.nanmax(level_ll))
    weights[np.isnan(weights)] = 0.0
    cdf = (weights / weights.sum()).cumsum()
    cdf /= cdf[-1]
    return cdf.searchsorted(uniforms, side="right").astype(np.int32)


def settle_near_ties(ties, evidences, sample_ll, base, uniforms, sample_z, sample_nhi, min_z_separation: float, level_eval):
    """
    Makes a device sweep's `base_sample_inds` np.random.choice's by construction.  The library's cdf and NumPy's differ by
    rounding (~1e-16), so the two can disagree on an index only where a uniform lies that close to a cdf step; the library
    counts, per sightline and level, the searches that ended within 1e-11 of a step (`ties`, dlagp_batch_resample_ties).
    For those levels -- at most two in a thousand at S = 10 000 (2 S margin per search when every sample carries weight) -- the indices are recomputed here with NumPy's own arithmetic
    (`exact_choice`) from the level's likelihoods as the device returned them.  They have never differed outside tests
    that plant uniforms on the steps; if they do, the sightline's remaining levels are redone from that level on with the
    host half of every level in NumPy (`level_eval(b, z (S, kd), nhi (S, kd)) -> (S,)` evaluates on the device).

    Arrays of the whole batch, modified in place: evidences (nb, max_dlas), sample_ll (nb, S, max_dlas), base
    (nb, max_dlas - 1, S), uniforms (nb, max_dlas - 1, S), sample_z (nb, S), sample_nhi (S,) or (nb, S).
    :return: number of (sightline, level) pairs whose indices changed
    """
    from .dla_gp import level_parameters

    ties = np.asarray(ties)
    changed = 0
    nb, S, max_dlas = sample_ll.shape
    for b in np.flatnonzero(ties.any(axis=1)):
        nhi = sample_nhi if sample_nhi.ndim == 1 else sample_nhi[b]
        level = 0
        while level + 1 < max_dlas:
            if np.isnan(evidences[b, level]):
                break  # the sweep stopped here (reference dla_gp.py:200-206): nothing was resampled
            if ties[b, level]:
                exact = exact_choice(sample_ll[b, :, level], uniforms[b, level])
                if not np.array_equal(exact, base[b, level]):
                    # NumPy draws another index somewhere: this level's indices are NumPy's, and everything behind them is
                    # recomputed -- device evaluations, host half of the levels in NumPy (all of them "tied" from here on)
                    changed += 1
                    base[b, level] = exact
                    base[b, level + 1:] = 0
                    sample_ll[b, :, level + 1:] = np.nan
                    evidences[b, level + 1:] = np.nan
                    ties[b, level + 1:] = 1
                    z, n, idx = level_parameters(sample_z[b], nhi, base[b], level + 1)
                    sample_ll[b, :, level + 1] = np.asarray(level_eval(b, z, n), dtype=np.float64) - np.log(S)
                    column = sample_ll[b, :, level + 1]
                    too_close = np.any(np.diff(np.sort(sample_z[b][idx], axis=0), axis=0) < min_z_separation, axis=0)
                    column[too_close] = np.nan  # reference dla_gp.py:164-177
                    if np.isnan(column).all():
                        evidences[b, level + 1] = np.nan
                    else:
                        top = np.nanmax(column)
                        evidences[b, level + 1] = top + np.log(np.nanmean(np.exp(column - top))) - np.log(S) * (level + 1)
            level += 1
    return changed


class DeviceBatch:
    """Sightlines of one device evaluated by one launch sequence; arrays are [nb][S][kd]."""

    def __init__(self, sightlines: Sequence[DeviceSightline]):
        lib = _lib.load()
        self.sightlines = list(sightlines)
        nb = len(self.sightlines)
        arr = (ctypes.c_void_p * nb)(*[s._handle for s in self.sightlines])
        handle = ctypes.c_void_p()
        _lib.check(lib.dlagp_batch_create(ctypes.byref(handle), arr, nb))
        self._handle = handle
        self.nb = nb
        self._pending = None  # (S, max_dlas, ...) between model_evidences_start and model_evidences_finish
        self.last_ties = None  # near-tie counts of the last multi-level sweep, and how many levels `settle_near_ties` changed
        self.last_changed = 0

    def close(self) -> None:
        if getattr(self, "_handle", None):
            _lib.load().dlagp_batch_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def eval(self, z, nhi) -> np.ndarray:
        z = _lib.as_f64(z)
        nhi = _lib.as_f64(nhi)
        if z.shape != nhi.shape:
            raise ValueError("z and nhi must have the same shape")
        if z.ndim == 2:
            z, nhi = z[:, :, None], nhi[:, :, None]
        if z.ndim != 3 or z.shape[0] != self.nb:
            raise ValueError("samples must be (nb, S) or (nb, S, kd)")
        z, nhi = np.ascontiguousarray(z), np.ascontiguousarray(nhi)
        _, S, kd = z.shape
        out = np.empty((self.nb, S), dtype=np.float64)
        _lib.check(_lib.load().dlagp_batch_eval(self._handle, S, kd, _lib.ptr(z), _lib.ptr(nhi), _lib.ptr(out)))
        return out

    def eval_device(self, S: int, kd: int, d_z: int, d_nhi: int, d_out: int, stream: int = 0) -> None:
        _lib.check(_lib.load().dlagp_batch_eval_device(self._handle, S, kd, d_z, d_nhi, d_out, stream))

    def eval_evidence_device(self, S: int, kd: int, d_z: int, d_nhi: int, d_out: int, level: int, d_evidence: int,
                             d_too_close: int = 0, stream: int = 0) -> None:
        """every member's level in one call: log-likelihoods minus log S and evidences (dlagp_batch_eval_evidence_device)"""
        _lib.check(_lib.load().dlagp_batch_eval_evidence_device(self._handle, S, kd, d_z, d_nhi, d_out, level,
                                                                d_too_close or None, d_evidence, stream))

    def resample_ties(self, max_dlas: int) -> np.ndarray:
        """(nb, max_dlas - 1) near-tie counts of the batch's most recent sweep (dlagp_batch_resample_ties)"""
        out = np.zeros((self.nb, max(max_dlas - 1, 0)), dtype=np.int32)
        if out.size:
            _lib.check(_lib.load().dlagp_batch_resample_ties(
                self._handle, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), out.size))
        return out

    def _settle(self, evidences, sample_ll, base, uniforms, sample_z, sample_nhi, min_z_separation):
        """`settle_near_ties` for a sweep this batch has just run; returns the (possibly copied) result arrays"""
        max_dlas = evidences.shape[1]
        if max_dlas < 2:
            return evidences, sample_ll, base
        ties = self.resample_ties(max_dlas)
        self.last_ties, self.last_changed = ties, 0
        if ties.any():
            self.last_changed = settle_near_ties(
                ties, evidences, sample_ll, base, uniforms, sample_z, sample_nhi, min_z_separation,
                lambda b, z, n: self.sightlines[b].eval(z, n))
        return evidences, sample_ll, base

    def nulls(self) -> np.ndarray:
        """null-model evidence of every member (NullGP.log_model_evidence, null_gp.py:294-305), one launch"""
        out = np.empty(self.nb, dtype=np.float64)
        _lib.check(_lib.load().dlagp_batch_null(self._handle, _lib.ptr(out)))
        return out


    def sweep(self, sample_z, sample_nhi, max_dlas: int, min_z_separation: float, uniforms=None):
        """
        Occupation sweep of every sightline with all levels on the device (reference dla_gp.py:92-225).

        :param sample_z: (nb, S) QMC redshifts;  sample_nhi: (S,) shared or (nb, S)
        :param uniforms: (nb, max_dlas - 1, S) U[0,1) draws that stand in for the reference's
            `np.random.choice` (see `parallel.resampling_uniforms`)
        :return: evidences (nb, max_dlas), sample_log_likelihoods (nb, S, max_dlas),
            base_sample_inds (nb, max_dlas - 1, S) int32 -- the reference's per-spectrum arrays, stacked
        """
        sample_z = np.ascontiguousarray(_lib.as_f64(sample_z))
        sample_nhi = np.ascontiguousarray(_lib.as_f64(sample_nhi))
        if sample_z.ndim != 2 or sample_z.shape[0] != self.nb:
            raise ValueError("sample_z must be (nb, S)")
        S = sample_z.shape[1]
        if sample_nhi.shape == (S,):
            stride = 0
        elif sample_nhi.shape == (self.nb, S):
            stride = S
        else:
            raise ValueError("sample_nhi must be (S,) or (nb, S)")
        max_dlas = int(max_dlas)
        if max_dlas > 1:
            if uniforms is None:
                raise ValueError("uniforms are required when max_dlas > 1")
            uniforms = np.ascontiguousarray(_lib.as_f64(uniforms))
            if uniforms.shape != (self.nb, max_dlas - 1, S):
                raise ValueError("uniforms must be (nb, max_dlas - 1, S)")
        evidences = np.empty((self.nb, max_dlas), dtype=np.float64)
        sample_ll = np.empty((self.nb, max_dlas, S), dtype=np.float64)
        base = np.zeros((self.nb, max(max_dlas - 1, 0), S), dtype=np.int32)
        _lib.check(_lib.load().dlagp_batch_sweep(
            self._handle, S, max_dlas, _lib.ptr(sample_z), _lib.ptr(sample_nhi), stride, float(min_z_separation),
            _lib.ptr(uniforms) if max_dlas > 1 else None, _lib.ptr(evidences), _lib.ptr(sample_ll),
            base.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)) if max_dlas > 1 else None))
        # (nb, S, max_dlas) as a VIEW of the level-major buffer the library fills: a contiguous copy costs 50 us per
        # sightline, and the consumers' reductions over the samples (np.nanargmax(..., axis=0) of the MAP estimate)
        # run ten times faster along the buffer's contiguous axis
        return self._settle(evidences, sample_ll.transpose(0, 2, 1), base, uniforms, sample_z, sample_nhi, min_z_separation)

    def model_evidences(self, sub_sample_z, sub_sample_nhi, sample_z, sample_nhi, max_dlas: int,
                        min_z_separation: float, uniforms=None):
        """
        The evidence loop of `BayesModelSelect.model_selection` (reference bayesian_model_selection.py:84-95) for the
        whole batch in one call into the library (dlagp_batch_model_evidences): `nulls()`, `sweep(sub_sample_z,
        sub_sample_nhi, 1, ...)` and `sweep(sample_z, sample_nhi, max_dlas, ...)` enqueued back to back and waited for
        once -- the same numbers, bit for bit.

        :return: null evidences (nb,), then the two `sweep` triples of the sub-DLA and the DLA model
        """
        nb = self.nb

        def tables(z, nhi):
            z = np.ascontiguousarray(_lib.as_f64(z))
            nhi = np.ascontiguousarray(_lib.as_f64(nhi))
            if z.ndim != 2 or z.shape[0] != nb:
                raise ValueError("sample_z must be (nb, S)")
            if nhi.shape == (z.shape[1],):
                return z, nhi, 0
            if nhi.shape == z.shape:
                return z, nhi, z.shape[1]
            raise ValueError("sample_nhi must be (S,) or (nb, S)")

        sub_z, sub_nhi, sub_stride = tables(sub_sample_z, sub_sample_nhi)
        z, nhi, stride = tables(sample_z, sample_nhi)
        S = z.shape[1]
        if sub_z.shape[1] != S:
            raise ValueError("both models need the same number of samples")
        max_dlas = int(max_dlas)
        if max_dlas > 1:
            if uniforms is None:
                raise ValueError("uniforms are required when max_dlas > 1")
            uniforms = np.ascontiguousarray(_lib.as_f64(uniforms))
            if uniforms.shape != (nb, max_dlas - 1, S):
                raise ValueError("uniforms must be (nb, max_dlas - 1, S)")
        nulls = np.empty(nb, dtype=np.float64)
        sub_ev = np.empty((nb, 1), dtype=np.float64)
        sub_ll = np.empty((nb, 1, S), dtype=np.float64)
        evidences = np.empty((nb, max_dlas), dtype=np.float64)
        sample_ll = np.empty((nb, max_dlas, S), dtype=np.float64)
        base = np.zeros((nb, max(max_dlas - 1, 0), S), dtype=np.int32)
        _lib.check(_lib.load().dlagp_batch_model_evidences(
            self._handle, S, max_dlas, _lib.ptr(sub_z), _lib.ptr(sub_nhi), sub_stride, _lib.ptr(z), _lib.ptr(nhi), stride,
            float(min_z_separation), _lib.ptr(uniforms) if max_dlas > 1 else None, _lib.ptr(nulls), _lib.ptr(sub_ev),
            _lib.ptr(sub_ll), _lib.ptr(evidences), _lib.ptr(sample_ll),
            base.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)) if max_dlas > 1 else None))
        return (nulls, (sub_ev, sub_ll.transpose(0, 2, 1), np.zeros((nb, 0, S), dtype=np.int32)),
                self._settle(evidences, sample_ll.transpose(0, 2, 1), base, uniforms, z, nhi, min_z_separation))

    def model_evidences_start(self, sub_sample_z, sub_sample_nhi, sample_z, sample_nhi, max_dlas: int,
                              min_z_separation: float) -> None:
        """
        First phase of `model_evidences` (dlagp_batch_model_evidences_start): the samples go up and the sub-DLA sweep,
        which needs no random numbers, starts; returns at once.  The caller draws the occupation sweep's uniforms and
        does its other host work while the device runs, then calls `model_evidences_finish`.
        """
        nb = self.nb

        def tables(z, nhi):
            z = np.ascontiguousarray(_lib.as_f64(z))
            nhi = np.ascontiguousarray(_lib.as_f64(nhi))
            if z.ndim != 2 or z.shape[0] != nb:
                raise ValueError("sample_z must be (nb, S)")
            if nhi.shape == (z.shape[1],):
                return z, nhi, 0
            if nhi.shape == z.shape:
                return z, nhi, z.shape[1]
            raise ValueError("sample_nhi must be (S,) or (nb, S)")

        sub_z, sub_nhi, sub_stride = tables(sub_sample_z, sub_sample_nhi)
        z, nhi, stride = tables(sample_z, sample_nhi)
        S = z.shape[1]
        if sub_z.shape[1] != S:
            raise ValueError("both models need the same number of samples")
        _lib.check(_lib.load().dlagp_batch_model_evidences_start(
            self._handle, S, int(max_dlas), _lib.ptr(sub_z), _lib.ptr(sub_nhi), sub_stride, _lib.ptr(z), _lib.ptr(nhi), stride,
            float(min_z_separation)))
        self._pending = (S, int(max_dlas), z, nhi, float(min_z_separation))

    def model_evidences_finish(self, uniforms=None):
        """second phase: same return value as `model_evidences`"""
        if self._pending is None:
            raise ValueError("model_evidences_finish without a matching model_evidences_start")
        S, max_dlas, z, nhi, min_z_separation = self._pending
        self._pending = None
        nb = self.nb
        if max_dlas > 1:
            if uniforms is None:
                raise ValueError("uniforms are required when max_dlas > 1")
            uniforms = np.ascontiguousarray(_lib.as_f64(uniforms))
            if uniforms.shape != (nb, max_dlas - 1, S):
                raise ValueError("uniforms must be (nb, max_dlas - 1, S)")
        nulls = np.empty(nb, dtype=np.float64)
        sub_ev = np.empty((nb, 1), dtype=np.float64)
        sub_ll = np.empty((nb, 1, S), dtype=np.float64)
        evidences = np.empty((nb, max_dlas), dtype=np.float64)
        sample_ll = np.empty((nb, max_dlas, S), dtype=np.float64)
        base = np.zeros((nb, max(max_dlas - 1, 0), S), dtype=np.int32)
        _lib.check(_lib.load().dlagp_batch_model_evidences_finish(
            self._handle, _lib.ptr(uniforms) if max_dlas > 1 else None, _lib.ptr(nulls), _lib.ptr(sub_ev), _lib.ptr(sub_ll),
            _lib.ptr(evidences), _lib.ptr(sample_ll),
            base.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)) if max_dlas > 1 else None))
        return (nulls, (sub_ev, sub_ll.transpose(0, 2, 1), np.zeros((nb, 0, S), dtype=np.int32)),
                self._settle(evidences, sample_ll.transpose(0, 2, 1), base, uniforms, z, nhi, min_z_separation))

    def model_evidences_cancel(self) -> None:
        """give up a `model_evidences_start` that will not be finished (the caller's own work in between raised)"""
        self._pending = None
        if getattr(self, "_handle", None):
            _lib.check(_lib.load().dlagp_batch_model_evidences_cancel(self._handle))

    def sweep_device(self, S: int, max_dlas: int, d_sample_z: int, d_sample_nhi: int, nhi_stride: int,
                     min_z_separation: float, d_uniforms: int, d_evidences: int, d_sample_ll: int, d_base_inds: int,
                     stream: int = 0) -> None:
        _lib.check(_lib.load().dlagp_batch_sweep_device(
            self._handle, S, max_dlas, d_sample_z, d_sample_nhi, nhi_stride, float(min_z_separation),
            d_uniforms or None, d_evidences, d_sample_ll, d_base_inds or None, stream))


def _two_model_sweep_device(self, S: int, max_dlas: int, d_sample_z: int, d_sub_nhi: int, sub_nhi_stride: int, d_sample_nhi: int,
                            nhi_stride: int, min_z_separation: float, d_uniforms: int, d_sub_evidences: int, d_sub_sample_ll: int,
                            d_evidences: int, d_sample_ll: int, d_base_inds: int, stream: int = 0) -> None:
    """
    The sub-DLA model's one-absorber sweep and the DLA model's occupation sweep at the SAME redshift samples, device
    pointers, enqueued on `stream` (dlagp_batch_two_model_sweep_device): one launch for both one-absorber levels when the
    batch's shape allows (`get_option("last_sweep_paired")`).
    """
    _lib.check(_lib.load().dlagp_batch_two_model_sweep_device(
        self._handle, S, max_dlas, d_sample_z, d_sub_nhi, sub_nhi_stride, d_sample_nhi, nhi_stride, float(min_z_separation),
        d_uniforms or None, d_sub_evidences, d_sub_sample_ll, d_evidences, d_sample_ll, d_base_inds or None, stream))


DeviceBatch.two_model_sweep_device = _two_model_sweep_device


def log_mvnpdf_low_rank(y, mu, M, d, device: int = 0) -> float:
    """stateless log N(y; mu, M M' + diag(d)) on the GPU (reference null_gp.py:307-360)"""
    y, mu, d = (_lib.as_f64(a) for a in (y, mu, d))
    M = _lib.strided_f64(M)
    if M.ndim != 2:
        raise ValueError("M must be (n, k)")
    n, k = M.shape
    out = np.empty(1, dtype=np.float64)
    rc = _lib.load().dlagp_mvnpdf(n, k, _lib.ptr(y), _lib.ptr(mu), _lib.ptr(M), M.strides[0] // 8,
                                  M.strides[1] // 8, _lib.ptr(d), _lib.ptr(out), int(device))
    _lib.check(rc)
    return float(out[0])


def level_evidence_device(d_ll: int, nb: int, S: int, level: int, d_evidence: int, d_too_close: int = 0,
                          device: int = 0, stream: int = 0) -> None:
    """
    In-place `ll -= log S`, optional separation mask, and the level's log-mean-exp evidence per
    sightline (reference dla_gp.py:155-190), all on device pointers; enqueued on `stream`.
    """
    _lib.check(_lib.load().dlagp_level_evidence_device(d_ll, nb, S, level, d_too_close or None, d_evidence,
                                                       device, stream))


def setup_sync(device: int = 0) -> None:
    """wait for the calling thread's deferred context set-up work (`DeviceSightline.from_model(..., defer_sync=True)`)"""
    _lib.check(_lib.load().dlagp_setup_sync(int(device)))


def set_option(name: str, value: int) -> None:
    _lib.check(_lib.load().dlagp_set_option(name.encode(), int(value)))


def set_options_from_env(variable: str = "DLAGP_OPTIONS") -> dict:
    """tuning runs: DLAGP_OPTIONS="level_spread=0,plan_slots=1024" sets those options; returns what it set"""
    import os

    done = {}
    for item in filter(None, os.environ.get(variable, "").split(",")):
        name, _, value = item.partition("=")
        set_option(name.strip(), int(value))
        done[name.strip()] = int(value)
    return done


def get_option(name: str) -> int:
    """current value of an option, or of a read-only diagnostic of the last launch ("last_launch_path": 1 generic /
    2 MFMA kernel, "last_launch_items", "last_launch_parts", "last_sweep_cached")"""
    value = ctypes.c_int64()
    _lib.check(_lib.load().dlagp_get_option(name.encode(), ctypes.byref(value)))
    return int(value.value)


def timing_read(reset: bool = True):
    """(total accumulate-kernel milliseconds, launches) recorded while option 'timing' was on"""
    total = ctypes.c_double()
    launches = ctypes.c_int64()
    _lib.check(_lib.load().dlagp_timing_read(ctypes.byref(total), ctypes.byref(launches), int(reset)))
    return total.value, launches.value


def debug_voigt(form: int, line: int, velocity, device: int = 0) -> np.ndarray:
    """Voigt profile of series member `line` at `velocity` [cm/s] as the kernels evaluate it (0 generic, 1 MFMA)"""
    velocity = np.ascontiguousarray(_lib.as_f64(velocity)).ravel()
    out = np.empty_like(velocity)
    _lib.check(_lib.load().dlagp_debug_voigt(int(form), int(line), velocity.shape[0], _lib.ptr(velocity),
                                             _lib.ptr(out), int(device)))
    return out


def debug_read(reset: bool = True):
    """per-phase shader-clock totals recorded while option 'debug' bit 1 was set (diagnostics)"""
    buf = (ctypes.c_uint64 * 8)()
    _lib.check(_lib.load().dlagp_debug_read(buf, 8, int(reset)))
    return list(buf)
