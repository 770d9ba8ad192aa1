import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def inputs():
    from greb_climate_model_amd import workload
    return workload.make_inputs()


@pytest.fixture(scope="session")
def params():
    """Reference defaults + the shipped namelist's diagnostic point (namelist:4-5)."""
    from greb_climate_model_amd import abi
    return abi.default_params(ipx=95, ipy=38)


@pytest.fixture(scope="session")
def oracle_lib():
    """Build (if needed) the C restatement.  The reference build (_ref) is NOT required by tests."""
    from oracle import oracle as O
    O.build(ref=False)
    return O


@pytest.fixture(scope="session")
def routine_golden():
    return load_golden("routine_g96.npz")


def rms(a, b):
    d = np.asarray(a, np.float64) - np.asarray(b, np.float64)
    return float(np.sqrt(np.mean(d * d)))
