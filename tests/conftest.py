import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]

    return load


def relerr(got, want):
    import numpy as np

    got = np.asarray(got, dtype=float)
    want = np.asarray(want, dtype=float)
    scale = max(float(np.abs(want).max()), 1e-300)
    return float(np.abs(got - want).max()) / scale


def relerr_elem(got, want, floor_frac=0.1):
    """Largest ELEMENT-WISE relative error, |got - want| / (|want| + floor_frac * rms(want)): unlike `relerr` (max-norm
    over max-norm) the small entries of an array are held to their own size, down to a floor of a tenth of the
    array's r.m.s. value (an entry crossing zero has no relative accuracy to speak of)."""
    import numpy as np

    got = np.asarray(got, dtype=float)
    want = np.asarray(want, dtype=float)
    floor = floor_frac * float(np.sqrt(np.mean(np.square(want)))) + 1e-300
    return float(np.max(np.abs(got - want) / (np.abs(want) + floor)))
