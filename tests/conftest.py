import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The library's launch deadline is opt-in (a plan() of the reference has no wall-clock limit); the suite gives every search launch two
# minutes so that a kernel bug ends a test with MPLX_ERR_TIMEOUT instead of wedging the run (tests/test_guard.py sets its own).
os.environ.setdefault("MPLX_DEADLINE_S", "120")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver)")


@pytest.fixture(scope="session")
def skir():
    import numpy as np
    d = np.load(os.path.join(ROOT, "tests", "golden", "skir_map.npz"))
    return d["grid"], d["origin"], float(d["res"])


@pytest.fixture(autouse=True)
def _release_device_contexts():
    """Device contexts own tens of GB of HBM in the scale tests; whatever a test leaves in a reference cycle is
    collected before the next test allocates."""
    yield
    import gc
    gc.collect()
