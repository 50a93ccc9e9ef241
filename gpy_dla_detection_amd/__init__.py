"""
MI355X-native GP model-evidence sweep for DLA detection.

Drop-in classes for the model-evidence path of jibanCat/gpy_dla_detection
(`NullGP.log_model_evidence`, `DLAGP/SubDLAGP.log_model_evidences`), evaluated by
hand-written FP64 HIP kernels for gfx950 behind a C-ABI shared library
(`csrc/libdlagp_hip.so`, declared in include/dlagp.h).

Importing the package does not load the HIP library; the first evaluation does,
and raises if the library is missing (there is no CPU fallback).
"""
__all__ = [
    "lyman_series",
    "set_parameters",
    "effective_optical_depth",
    "samples",
    "synthetic",
]
