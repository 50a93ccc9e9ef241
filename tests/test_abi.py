"""
CPU-side checks of the C-ABI boundary: the shared library loads, exports every symbol
include/dlagp.h declares, and the compiled constants equal the Python tables.  No compute calls.
"""
import ctypes
import os
import re

import numpy as np
import pytest

from gpy_dla_detection_amd import _lib, lyman_series as ls

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "dlagp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dlagp_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build with `make -C gpy_dla_detection_amd/csrc`"
    assert _lib.load().dlagp_abi_version() == _lib.ABI_VERSION == 3  # first: it maps the one HIP runtime this process is to have
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_lyman_header_matches_python_tables():
    text = open(os.path.join(REPO, "include", "dlagp_lyman.h")).read()

    def table(name):
        body = re.search(name + r"\[\d+\] = \{(.*?)\};", text, flags=re.S).group(1)
        return np.array([float.fromhex(tok) for tok in body.replace("\n", " ").split(",") if tok.strip()])

    np.testing.assert_array_equal(table("DLAGP_TRANSITION_WAVELENGTHS_CM"), ls.TRANSITION_WAVELENGTHS)
    np.testing.assert_array_equal(table("DLAGP_LEADING_CONSTANTS"), ls.LEADING_CONSTANTS)
    np.testing.assert_array_equal(table("DLAGP_LORENTZ_WIDTHS"), ls.LORENTZ_WIDTHS)
    np.testing.assert_array_equal(table("DLAGP_INSTRUMENT_PROFILE"), ls.INSTRUMENT_PROFILE)
    c = float.fromhex(re.search(r"DLAGP_SPEED_OF_LIGHT_CGS (\S+)", text).group(1))
    sigma = float.fromhex(re.search(r"DLAGP_SIGMA_CGS (\S+)", text).group(1))
    assert c == ls.SPEED_OF_LIGHT_CGS and sigma == ls.SIGMA_CGS


def test_argument_errors_do_not_need_a_gpu():
    lib = _lib.load()
    # NULL context / bad kd are rejected before any device call
    out = np.zeros(1)
    assert lib.dlagp_eval(None, 1, 1, _lib.ptr(out), _lib.ptr(out), _lib.ptr(out)) == _lib.DLAGP_EINVAL
    assert "NULL" in _lib.last_error()
    assert lib.dlagp_set_option(b"no_such_option", 1) == _lib.DLAGP_EINVAL
    with pytest.raises(ValueError):
        _lib.check(_lib.DLAGP_EINVAL)
    with pytest.raises(np.linalg.LinAlgError):
        _lib.check(_lib.DLAGP_ELINALG)
    with pytest.raises(RuntimeError):
        _lib.check(_lib.DLAGP_EDEVICE)


def test_cpp_exceptions_come_back_as_status_codes():
    """
    include/dlagp.h: "Nothing aborts the process".  Every exported function is an exception barrier
    (dlagp_hip.hip exception_status): a std::bad_alloc raised inside the library must come back through ctypes as
    DLAGP_ENOMEM with a message, not unwind into the caller and end in std::terminate.  Forced here without a GPU and
    without touching memory: the planner is asked for the piece counts of 2^58 blocks of samples (one vector of 2^58
    bytes, which no allocator grants).  In a child process, so that a missing barrier reads as a failed test.
    """
    import subprocess
    import sys

    code = """
import ctypes, numpy as np
from gpy_dla_detection_amd import _lib
lib = _lib.load()
keep = np.ones(64, dtype=np.uint8)
n_u = (ctypes.c_int * 1)(64)
keeps = (ctypes.POINTER(ctypes.c_ubyte) * 1)(keep.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)))
summary = (ctypes.c_int64 * 8)()
rc = lib.dlagp_plan_debug(1, n_u, keeps, 7, 2 ** 62, None, summary, None, None, 0, None, 0, None, 0, None, 0)
print(rc, _lib.last_error())
try:
    _lib.check(rc)
except MemoryError as exc:
    print("MemoryError")
# the library is still usable afterwards
rc = lib.dlagp_plan_debug(1, n_u, keeps, 7, 100, None, summary, None, None, 0, None, 0, None, 0, None, 0)
print(rc, summary[2])
"""
    res = subprocess.run([sys.executable, "-c", code], cwd=REPO, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, (res.returncode, res.stderr[-2000:])
    lines = res.stdout.strip().splitlines()
    assert lines[0].startswith("%d " % _lib.DLAGP_ENOMEM) and "dlagp_plan_debug" in lines[0] and "memory" in lines[0], lines
    assert lines[1] == "MemoryError"
    assert lines[2].split()[0] == "0" and int(lines[2].split()[1]) >= 7


def test_one_hip_runtime_per_process_whatever_the_import_order():
    """
    torch bundles a HIP runtime with the SONAME the library needs.  Whichever of the two is imported first, the process
    must end up with ONE copy mapped (`_lib._preload_torch_hip_runtime`): with two, streams and RCCL collectives of
    torch's runtime and the launches of libdlagp_hip.so are unordered with respect to each other.
    """
    import subprocess
    import sys

    pytest.importorskip("torch")
    for order in ("from gpy_dla_detection_amd import _lib; _lib.load(); import torch",
                  "import torch; from gpy_dla_detection_amd import _lib; _lib.load()"):
        code = order + "; print(len(_lib.hip_runtime_copies())); print(_lib.hip_runtime()._name)"
        res = subprocess.run([sys.executable, "-c", code], cwd=REPO, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr[-2000:]
        assert res.stdout.split()[:2] == ["1", "libamdhip64.so.7"], (order, res.stdout)
