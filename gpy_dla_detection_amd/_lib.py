"""
ctypes binding of libdlagp_hip.so (C ABI: include/dlagp.h).

This is the only place the shared library is touched.  There is no fallback: if the
library is missing or no GPU is usable, the first evaluation raises.
"""
import ctypes
import os
import sys
from ctypes import POINTER, c_char_p, c_double, c_int, c_int64, c_ssize_t, c_ubyte, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DLAGP_LIB lets a tuning session load an alternative build of the same ABI (A/B runs on one device)
LIB_PATH = os.environ.get("DLAGP_LIB") or os.path.join(_HERE, "csrc", "libdlagp_hip.so")

ABI_VERSION = 3            # dlagp_abi_version() this binding is written for
LEVEL_EXCHANGE = 2064      # DLAGP_LEVEL_EXCHANGE of include/dlagp.h: scratch doubles per sightline of dlagp_level_finish_device

# error codes of include/dlagp.h
DLAGP_OK, DLAGP_EINVAL, DLAGP_EDEVICE, DLAGP_ELINALG, DLAGP_ENOMEM = 0, -1, -2, -3, -4

# every symbol include/dlagp.h declares: name -> (restype, argtypes)
_dp = POINTER(c_double)
SIGNATURES = {
    "dlagp_abi_version": (c_int, []),
    "dlagp_device_count": (c_int, []),
    "dlagp_last_error": (c_char_p, []),
    "dlagp_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, _dp, _dp, _dp, c_ssize_t, c_ssize_t,
                             _dp, _dp, _dp, POINTER(c_ubyte), c_int, c_int, c_int]),
    "dlagp_destroy": (None, [c_void_p]),
    "dlagp_model_create": (c_int, [POINTER(c_void_p), c_int, c_int, _dp, _dp, _dp, c_ssize_t, c_ssize_t, _dp,
                                   c_double, c_double, c_double, c_double, c_double, c_int, c_int]),
    "dlagp_model_destroy": (None, [c_void_p]),
    "dlagp_create_from_model": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_int, _dp, _dp, _dp, _dp, c_double,
                                        _dp, POINTER(c_ubyte), c_int, c_int]),
    "dlagp_setup_sync": (c_int, [c_int]),
    "dlagp_ctx_read_model": (c_int, [c_void_p, _dp, _dp, _dp]),
    "dlagp_ctx_n": (c_int, [c_void_p]),
    "dlagp_ctx_k": (c_int, [c_void_p]),
    "dlagp_ctx_device": (c_int, [c_void_p]),
    "dlagp_eval": (c_int, [c_void_p, c_int64, c_int, _dp, _dp, _dp]),
    "dlagp_null": (c_int, [c_void_p, _dp]),
    "dlagp_mvnpdf": (c_int, [c_int, c_int, _dp, _dp, _dp, c_ssize_t, c_ssize_t, _dp, _dp, c_int]),
    "dlagp_absorption": (c_int, [c_void_p, c_int, _dp, _dp, _dp]),
    "dlagp_eval_device": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dlagp_batch_create": (c_int, [POINTER(c_void_p), POINTER(c_void_p), c_int]),
    "dlagp_batch_destroy": (None, [c_void_p]),
    "dlagp_batch_eval_device": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dlagp_batch_eval": (c_int, [c_void_p, c_int64, c_int, _dp, _dp, _dp]),
    "dlagp_batch_null": (c_int, [c_void_p, _dp]),
    "dlagp_batch_sweep": (c_int, [c_void_p, c_int64, c_int, _dp, _dp, c_int64, c_double, _dp, _dp, _dp,
                                  POINTER(ctypes.c_int32)]),
    "dlagp_batch_sweep_device": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, c_double, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p]),
    "dlagp_batch_two_model_sweep_device": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                                   c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                   c_void_p]),
    "dlagp_batch_model_evidences": (c_int, [c_void_p, c_int64, c_int, _dp, _dp, c_int64, _dp, _dp, c_int64, c_double,
                                            _dp, _dp, _dp, _dp, _dp, _dp, POINTER(ctypes.c_int32)]),
    "dlagp_batch_model_evidences_start": (c_int, [c_void_p, c_int64, c_int, _dp, _dp, c_int64, _dp, _dp, c_int64, c_double]),
    "dlagp_batch_model_evidences_finish": (c_int, [c_void_p, _dp, _dp, _dp, _dp, _dp, _dp, POINTER(ctypes.c_int32)]),
    "dlagp_batch_model_evidences_cancel": (c_int, [c_void_p]),
    "dlagp_batch_resample_ties": (c_int, [c_void_p, POINTER(ctypes.c_int32), c_int64]),
    "dlagp_batch_level_eval_device": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int64,
                                              c_void_p, c_int, c_void_p, c_void_p]),
    "dlagp_sweep_state_init_device": (c_int, [c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                              c_void_p]),
    "dlagp_level_finish_device": (c_int, [c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_double, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "dlagp_level_evidence_device": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "dlagp_eval_evidence_device": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "dlagp_batch_eval_evidence_device": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                                 c_void_p]),
    "dlagp_set_option": (c_int, [c_char_p, c_int64]),
    "dlagp_plan_cut": (c_int, [c_int64, c_double, c_int64, POINTER(c_int), _dp]),
    "dlagp_plan_debug": (c_int, [c_int, POINTER(c_int), POINTER(POINTER(ctypes.c_ubyte)), c_int, c_int64,
                                 POINTER(c_int64), POINTER(c_int64), POINTER(ctypes.c_int32), POINTER(ctypes.c_int32),
                                 c_int64, POINTER(ctypes.c_int32), c_int64, POINTER(ctypes.c_int32), c_int64,
                                 POINTER(ctypes.c_ubyte), c_int64]),
    "dlagp_get_option": (c_int, [c_char_p, POINTER(c_int64)]),
    "dlagp_timing_read": (c_int, [_dp, POINTER(c_int64), c_int]),
    "dlagp_debug_voigt": (c_int, [c_int, c_int, c_int64, _dp, _dp, c_int]),
    "dlagp_debug_read": (c_int, [POINTER(ctypes.c_uint64), c_int, c_int]),
}



class PrepareParams(ctypes.Structure):
    """dlagp_prepare_params of include/dlagp.h"""
    _fields_ = [(name, c_double) for name in (
        "min_lambda", "max_lambda", "normalization_min_lambda", "normalization_max_lambda", "pixel_spacing",
        "lya_wavelength", "lyman_limit", "min_z_cut", "max_z_cut")] + [("width", ctypes.c_int32), ("normalize", ctypes.c_int32)]


# dlagp_pad_callback: (user, nb, width, pixel_spacing, edges [nb][2], pad [nb][2 width])
PAD_CALLBACK = ctypes.CFUNCTYPE(None, c_void_p, ctypes.c_int32, ctypes.c_int32, c_double, _dp, _dp)
_pp = POINTER(c_void_p)  # arrays of array pointers
SIGNATURES["dlagp_prepare_spectrum"] = (c_int, [POINTER(PrepareParams), c_int, _dp, _dp, _dp, POINTER(c_ubyte), c_double, _dp,
                                                POINTER(c_int), POINTER(c_int), _dp, _dp, _dp, _dp, _dp, _dp, _dp,
                                                POINTER(c_ubyte)])
SIGNATURES["dlagp_batch_prepare"] = (c_int, [POINTER(c_void_p), POINTER(c_void_p), c_void_p, POINTER(PrepareParams), c_int,
                                             POINTER(c_int), _pp, _pp, _pp, _pp, _dp, c_int, c_int, PAD_CALLBACK, c_void_p,
                                             c_int64, _dp, _dp, _dp, _dp, _dp])

_lib = None
_hip = None


def _preload_torch_hip_runtime() -> None:
    """
    One HIP runtime per process.  libdlagp_hip.so needs `libamdhip64.so.7`; PyTorch-ROCm bundles a runtime of its own
    with that same SONAME (torch/lib/libamdhip64.so).  The dynamic loader binds the library to whichever copy is
    already in the process: torch's when torch was imported first -- one runtime, torch streams and tensors can be
    handed to the `*_device` entry points, RCCL and our kernels are ordered on the same streams -- but the system's
    when the library was loaded first, after which `import torch` brings in a SECOND runtime (separate streams, no
    ordering between them, and on this stack the second runtime to initialise does not see the GPU).  So whenever
    torch is installed its runtime is mapped first, without importing torch.  DLAGP_HIP_RUNTIME=system opts out
    (a process that never touches torch).
    """
    if os.environ.get("DLAGP_HIP_RUNTIME", "").lower() == "system" or "torch" in sys.modules:
        return
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        # only a runtime of the SONAME the library was linked against can stand in for it: a torch wheel built for
        # another ROCm major is left alone (the library then binds to the system runtime, and `hip_runtime_copies`
        # reports two runtimes once torch is imported)
        wanted = _needed_hip_soname()
        if os.path.exists(path) and wanted and _file_contains(path, wanted.encode() + b"\0"):
            ctypes.CDLL(path)  # RTLD_LOCAL: the loader's SONAME match needs no global symbols
    except Exception:  # no torch, a CPU-only torch, an unreadable install: the system runtime serves
        pass


def _file_contains(path: str, needle: bytes) -> bool:
    import mmap

    with open(path, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as view:
        return view.find(needle) >= 0


def _needed_hip_soname() -> str:
    """the `libamdhip64.so.N` libdlagp_hip.so was linked against (its DT_NEEDED entry), '' if it cannot be read"""
    import mmap
    import re

    try:
        with open(LIB_PATH, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as view:
            found = re.search(rb"libamdhip64\.so\.\d+", view)
            return found.group(0).decode() if found else ""
    except OSError:
        return ""


def load():
    """Load the HIP library once; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libdlagp_hip.so not found at {} -- build it with `make -C gpy_dla_detection_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback".format(LIB_PATH)
            )
        _preload_torch_hip_runtime()
        lib = ctypes.CDLL(LIB_PATH)
        copies = hip_runtime_copies()
        if len(copies) > 1:  # e.g. the system runtime was mapped before this module had a say
            import warnings

            warnings.warn("two HIP runtimes are mapped into this process (%s): streams and tensors of one are not valid "
                          "in the other; import gpy_dla_detection_amd before anything that loads the system runtime, "
                          "or set DLAGP_HIP_RUNTIME=system in a process that never imports torch" % ", ".join(copies),
                          RuntimeWarning, stacklevel=2)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.dlagp_abi_version() != ABI_VERSION:
            raise RuntimeError("{} has ABI version {}, this binding is written for {}: rebuild it "
                               "(`make -C gpy_dla_detection_amd/csrc`)".format(LIB_PATH, lib.dlagp_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


def hip_runtime():
    """ctypes handle of the HIP runtime the library is bound to (tests allocate device memory through it)"""
    global _hip
    if _hip is None:
        load()
        _hip = ctypes.CDLL("libamdhip64.so.7")  # SONAME lookup: the copy already mapped, never a second one
    return _hip


def hip_runtime_copies():
    """paths of the libamdhip64 copies mapped into this process (diagnostic: more than one is a set-up error)"""
    seen = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.split()[-1]
                if "libamdhip64" in path and path not in seen:
                    seen.append(path)
    except OSError:
        pass
    return seen


def last_error() -> str:
    return load().dlagp_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    """Map a C status to the exception the reference would raise at that point."""
    if rc == DLAGP_OK:
        return
    msg = last_error()
    if rc == DLAGP_EINVAL:
        raise ValueError(msg)
    if rc == DLAGP_ELINALG:
        raise np.linalg.LinAlgError(msg)  # numpy.linalg.cholesky at reference null_gp.py:343
    if rc == DLAGP_ENOMEM:
        raise MemoryError(msg)  # what NumPy raises when an allocation of the reference's fails
    raise RuntimeError(msg)


def as_f64(a) -> np.ndarray:
    """contiguous float64 view/copy (the reference's tests pass integer arrays, tests/test_model.py:53-55)"""
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a: np.ndarray):
    return a.ctypes.data_as(_dp)


def strided_f64(M) -> np.ndarray:
    """float64 array usable with element strides (Fortran order is passed through without a copy)"""
    M = np.asarray(M)
    if M.dtype != np.float64 or any(s % 8 for s in M.strides) or any(s < 0 for s in M.strides):
        M = np.array(M, dtype=np.float64, order="C")
    return M
