"""
Pipeline parameters consumed by the GP classes of this package.

Mirrors the attribute surface of the reference `Parameters` object
(reference: gpy_dla_detection/set_parameters.py:14-165) so that scripts written
against it (run_bayes_select.py:81) keep working: same constructor keywords,
same attribute names, same helper methods.  Only the attributes read on the
model-evidence path matter to the kernels: `k`, `num_dla_samples`, `num_lines`,
`width`, `pixel_spacing`, `min_lambda`, `max_lambda` and the z-range helpers.
"""
from typing import Optional

import numpy as np


class Parameters:
    # physical constants (class attributes, as in the reference :16-19)
    lya_wavelength: float = 1215.6701   # Å
    lyb_wavelength: float = 1025.7223   # Å
    lyman_limit: float = 911.7633       # Å
    speed_of_light: float = 299792458.0  # m s^-1

    _DEFAULTS = dict(
        loading_min_lambda=910.0,
        loading_max_lambda=1217.0,
        z_qso_cut=2.15,
        min_num_pixels=200,
        normalization_min_lambda=1310.0,
        normalization_max_lambda=1325.0,
        min_lambda=911.75,
        max_lambda=1215.75,
        dlambda=0.25,
        k=20,
        max_noise_variance=3.0 ** 2,
        initial_c_0=0.1,
        initial_tau_0=0.0023,
        initial_beta=3.65,
        minFunc_options=None,
        num_dla_samples=10000,
        alpha=0.97,
        uniform_min_log_nhi=20.0,
        uniform_max_log_nhi=23.0,
        fit_min_log_nhi=20.0,
        fit_max_log_nhi=22.0,
        prior_z_qso_increase=30000.0,
        width=3,
        pixel_spacing=1e-4,
        num_lines=3,
        max_z_cut=3000.0,
        min_z_cut=3000.0,
        num_forest_lines=31,
    )

    # keywords given in km/s and stored as redshift differences
    _KMS_FIELDS = ("prior_z_qso_increase", "max_z_cut", "min_z_cut")

    def __init__(self, *args, **kwargs):
        # positional arguments follow the reference's keyword order (:21-66)
        names = list(self._DEFAULTS)
        if len(args) > len(names):
            raise TypeError("too many positional parameters")
        for name, value in zip(names, args):
            if name in kwargs:
                raise TypeError("parameter {} given twice".format(name))
            kwargs[name] = value
        unknown = set(kwargs) - set(self._DEFAULTS)
        if unknown:
            raise TypeError("unexpected parameter(s): {}".format(sorted(unknown)))
        for name, default in self._DEFAULTS.items():
            value = kwargs.get(name, default)
            if name == "minFunc_options" and value is None:
                value = {"MaxIter": 2000, "MaxFunEvals": 4000}
            if name in self._KMS_FIELDS:
                value = self.kms_to_z(value)
            setattr(self, name, value)

    @classmethod
    def kms_to_z(cls, kms: float) -> float:
        """relative velocity in km/s -> redshift difference (reference :105-110)"""
        return (kms * 1000) / cls.speed_of_light

    @staticmethod
    def emitted_wavelengths(observed_wavelengths: np.ndarray, z: float) -> np.ndarray:
        return observed_wavelengths / (1 + z)

    @staticmethod
    def observed_wavelengths(emitted_wavelengths: np.ndarray, z: float) -> np.ndarray:
        return emitted_wavelengths * (1 + z)

    def _in_model_range(self, wavelengths: np.ndarray, z_qso: float) -> np.ndarray:
        rest = self.emitted_wavelengths(wavelengths, z_qso)
        return (rest >= self.min_lambda) & (rest <= self.max_lambda)

    def max_z_dla(self, wavelengths: np.ndarray, z_qso: float) -> float:
        """largest absorber redshift searched (reference :125-140)"""
        ind = self._in_model_range(wavelengths, z_qso)
        return np.min(
            [
                (np.max(wavelengths[ind]) / self.lya_wavelength - 1) - self.max_z_cut,
                z_qso - self.max_z_cut,
            ]
        )

    def min_z_dla(self, wavelengths: np.ndarray, z_qso: float) -> float:
        """smallest absorber redshift searched (reference :142-159)"""
        ind = self._in_model_range(wavelengths, z_qso)
        return np.max(
            [
                np.min(wavelengths[ind]) / self.lya_wavelength - 1,
                self.observed_wavelengths(self.lyman_limit, z_qso) / self.lya_wavelength
                - 1
                + self.min_z_cut,
            ]
        )

    def z_dla_range(self, wavelengths: np.ndarray, z_qso: float):
        """(`min_z_dla`, `max_z_dla`) with the in-range test made once: the same expressions, the same values"""
        ind = self._in_model_range(wavelengths, z_qso)
        inside = wavelengths[ind]
        z_min = np.max(
            [
                np.min(inside) / self.lya_wavelength - 1,
                self.observed_wavelengths(self.lyman_limit, z_qso) / self.lya_wavelength - 1 + self.min_z_cut,
            ]
        )
        z_max = np.min([(np.max(inside) / self.lya_wavelength - 1) - self.max_z_cut, z_qso - self.max_z_cut])
        return z_min, z_max

    def __repr__(self) -> str:
        return str(self.__dict__)
