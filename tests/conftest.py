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
