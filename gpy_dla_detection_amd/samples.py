"""
Quasi-Monte-Carlo parameter samples (z_DLA offset, log N_HI) for the absorber models.

Array-backed holders with the attribute surface the GP classes read
(reference: gpy_dla_detection/dla_samples.py:52-104 `DLASamplesMAT`,
subdla_samples.py:66-125 `SubDLASamplesMAT`): `offset_samples`,
`log_nhi_samples`, `nhi_samples`, `sample_z_dlas()` / `sample_z_lls()`,
and for the sub-DLA set `Z_dla` / `Z_lls`.

The reference loads these arrays from MATLAB v7.3 files through h5py; the
`from_mat` constructors do the same when h5py is installed.  `halton_*`
builds the synthetic sample set of SURVEY.md §8(d) used by bench.py and tests.
"""
from typing import Optional

import numpy as np

from .set_parameters import Parameters


def halton(num: int, base: int) -> np.ndarray:
    """First `num` points (index 1..num) of the unscrambled van der Corput sequence."""
    idx = np.arange(1, num + 1, dtype=np.int64)
    out = np.zeros(num, dtype=np.float64)
    f = 1.0
    while np.any(idx > 0):
        f = f / base
        out += f * (idx % base)
        idx //= base
    return out


class DLASamples:
    """
    QMC samples theta = (z_dla, log N_HI) of the DLA model.

    :param params: pipeline parameters (z-range helpers are read from it)
    :param prior: prior catalogue, kept for signature compatibility (may be None)
    :param offset_samples: (S,) offsets in [0, 1] mapped onto [min_z_dla, max_z_dla]
    :param log_nhi_samples: (S,) log10 column densities
    :param nhi_samples: (S,) column densities; default 10 ** log_nhi_samples
    """

    def __init__(
        self,
        params: Parameters,
        prior,
        offset_samples: np.ndarray,
        log_nhi_samples: np.ndarray,
        nhi_samples: Optional[np.ndarray] = None,
    ):
        self.params = params
        self.prior = prior

        self.num_dla_samples = params.num_dla_samples
        self.uniform_min_log_nhi = params.uniform_min_log_nhi
        self.uniform_max_log_nhi = params.uniform_max_log_nhi
        self.fit_min_log_nhi = params.fit_min_log_nhi
        self.fit_max_log_nhi = params.fit_max_log_nhi
        self.alpha = params.alpha

        self._offset_samples = np.ascontiguousarray(offset_samples, dtype=np.float64)
        self._log_nhi_samples = np.ascontiguousarray(log_nhi_samples, dtype=np.float64)
        if nhi_samples is None:
            nhi_samples = 10.0 ** self._log_nhi_samples
        self._nhi_samples = np.ascontiguousarray(nhi_samples, dtype=np.float64)
        assert self._offset_samples.shape == self._log_nhi_samples.shape
        assert self._offset_samples.shape == self._nhi_samples.shape

    @property
    def offset_samples(self) -> np.ndarray:
        return self._offset_samples

    @property
    def log_nhi_samples(self) -> np.ndarray:
        return self._log_nhi_samples

    @property
    def nhi_samples(self) -> np.ndarray:
        return self._nhi_samples

    def _map_offsets(self, wavelengths: np.ndarray, z_qso: float) -> np.ndarray:
        params = self.params
        if type(params).min_z_dla is Parameters.min_z_dla and type(params).max_z_dla is Parameters.max_z_dla:
            z_min, z_max = params.z_dla_range(wavelengths, z_qso)  # the two bounds with the in-range test made once
        else:  # a subclass with bounds of its own
            z_min = params.min_z_dla(wavelengths, z_qso)
            z_max = params.max_z_dla(wavelengths, z_qso)
        return z_min + (z_max - z_min) * self._offset_samples

    def sample_z_dlas(self, wavelengths: np.ndarray, z_qso: float) -> np.ndarray:
        """absorber redshifts for this sightline (reference dla_samples.py:94-104)"""
        return self._map_offsets(wavelengths, z_qso)

    @classmethod
    def from_mat(cls, params: Parameters, prior, dla_samples_file: str = "dla_samples_a03.mat"):
        """Load `generate_dla_samples.m` output (reference dla_samples.py:67-74)."""
        import h5py  # optional dependency

        with h5py.File(dla_samples_file, "r") as f:
            self = cls(
                params,
                prior,
                f["offset_samples"][:, 0],
                f["log_nhi_samples"][:, 0],
                f["nhi_samples"][:, 0],
            )
            # the file must have been generated with these parameters (reference dla_samples.py:69-70) ...
            assert self.alpha == f["alpha"][0, 0]
            assert self.uniform_min_log_nhi == f["uniform_min_log_nhi"][0, 0]
            # ... and its column-density range is the one the MCMC prior uses (reference :76-77)
            self.uniform_min_log_nhi = f["uniform_min_log_nhi"][0, 0]
            self.uniform_max_log_nhi = f["uniform_max_log_nhi"][0, 0]
        return self

    @classmethod
    def halton_synthetic(cls, params: Parameters, prior=None, num: Optional[int] = None):
        """offsets = Halton base 2, log N = 20 + 3 * Halton base 3."""
        num = params.num_dla_samples if num is None else num
        return cls(params, prior, halton(num, 2), 20.0 + 3.0 * halton(num, 3))


class SubDLASamples(DLASamples):
    """
    QMC samples of the sub-DLA model (log N_HI in [19.5, 20]).

    `Z_dla` / `Z_lls` are the partition functions used to rescale the sub-DLA
    model prior (reference subdla_samples.py:92-93, subdla_gp.py:311-346).
    """

    def __init__(
        self,
        params: Parameters,
        prior,
        offset_samples: np.ndarray,
        log_nhi_samples: np.ndarray,
        nhi_samples: Optional[np.ndarray] = None,
        Z_dla: float = 1.0,
        Z_lls: float = 1.0,
        extrapolate_min_log_nhi: float = 19.5,
    ):
        self.extrapolate_min_log_nhi = extrapolate_min_log_nhi
        super().__init__(params, prior, offset_samples, log_nhi_samples, nhi_samples)
        self._Z_dla = Z_dla
        self._Z_lls = Z_lls

    @property
    def Z_dla(self) -> float:
        return self._Z_dla

    @property
    def Z_lls(self) -> float:
        return self._Z_lls

    def sample_z_lls(self, wavelengths: np.ndarray, z_qso: float) -> np.ndarray:
        """absorber redshifts for this sightline (reference subdla_samples.py:115-125)"""
        return self._map_offsets(wavelengths, z_qso)

    @classmethod
    def from_mat(cls, params: Parameters, prior, sub_dla_samples_file: str = "subdla_samples.mat"):
        """Load `set_lls_parameters.m` output (reference subdla_samples.py:76-93)."""
        import h5py  # optional dependency

        with h5py.File(sub_dla_samples_file, "r") as f:
            self = cls(
                params,
                prior,
                f["offset_samples"][:, 0],
                f["lls_log_nhi_samples"][:, 0],
                f["lls_nhi_samples"][:, 0],
                Z_dla=f["Z_dla"][0, 0],
                Z_lls=f["Z_lls"][0, 0],
                extrapolate_min_log_nhi=f["extrapolate_min_log_nhi"][0, 0],
            )
            # consistency with the parameters the file was generated with (reference subdla_samples.py:83-84)
            assert self.alpha == f["alpha"][0, 0]
            assert self.num_dla_samples == f["num_dla_samples"][0, 0]
        return self

    @classmethod
    def halton_synthetic(cls, params: Parameters, prior=None, num: Optional[int] = None):
        """offsets = Halton base 2, log N = 19.5 + 0.5 * Halton base 3."""
        num = params.num_dla_samples if num is None else num
        return cls(params, prior, halton(num, 2), 19.5 + 0.5 * halton(num, 3))


# names used by reference-style scripts
DLASamplesMAT = DLASamples.from_mat
SubDLASamplesMAT = SubDLASamples.from_mat
