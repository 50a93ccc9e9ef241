import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


PARITY_REPORT = []  # lines the full-population comparisons leave for the end of the run (they are what a reviewer reads)


def report_parity(label: str, worst: float, count: int, note: str = "") -> None:
    PARITY_REPORT.append("parity %-58s worst relative difference to the oracle %.3e over %d evaluations%s"
                         % (label, worst, count, (" " + note) if note else ""))


def pytest_terminal_summary(terminalreporter):
    if PARITY_REPORT:
        terminalreporter.write_sep("-", "HIP path against the oracle, whole populations")
        for line in PARITY_REPORT:
            terminalreporter.write_line(line)


def free_port() -> int:
    """a TCP port nobody listens on right now (two test runs on one host must not share a rendezvous port)"""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def oracle_state(sight, num_lines=3, broadening=True):
    """SpectrumState of the oracle for a synthetic sightline"""
    from oracle import ref_like

    return ref_like.SpectrumState(
        y=sight.y,
        v=sight.v,
        this_mu=sight.this_mu,
        this_M=sight.this_M,
        this_omega2=sight.this_omega2,
        padded_wavelengths=sight.padded_wavelengths,
        unmasked_wavelengths=sight.unmasked_wavelengths,
        keep=sight.keep,
        num_lines=num_lines,
        broadening=broadening,
    )


class HipBuffers:
    """
    Minimal device-memory helper for tests that pass HBM pointers through the C ABI: hipMalloc /
    hipMemcpy from the HIP runtime the library itself is bound to (`_lib.hip_runtime()`: one runtime
    per process, see `_lib._preload_torch_hip_runtime`).
    """

    def __init__(self):
        import ctypes

        from gpy_dla_detection_amd import _lib

        self.ct = ctypes
        self.hip = _lib.hip_runtime()
        self.ptrs = []

    def to_device(self, array):
        arr = np.ascontiguousarray(array)
        ptr = self.ct.c_void_p()
        assert self.hip.hipMalloc(self.ct.byref(ptr), self.ct.c_size_t(max(arr.nbytes, 8))) == 0
        assert self.hip.hipMemcpy(ptr, arr.ctypes.data_as(self.ct.c_void_p), self.ct.c_size_t(arr.nbytes), 1) == 0
        self.ptrs.append(ptr)
        return ptr.value

    def to_host(self, ptr, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        assert self.hip.hipDeviceSynchronize() == 0
        assert self.hip.hipMemcpy(out.ctypes.data_as(self.ct.c_void_p), self.ct.c_void_p(ptr), self.ct.c_size_t(out.nbytes), 2) == 0
        return out

    def free(self):
        for ptr in self.ptrs:
            self.hip.hipFree(ptr)
        self.ptrs = []
