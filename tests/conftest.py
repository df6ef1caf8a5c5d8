import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """tests/golden/<name>.npz -> (GridMap, dict of arrays)."""
    from pyracecarsimulator_amd import maps
    z = dict(np.load(os.path.join(GOLD, name + ".npz")))
    rows, cols = (int(v) for v in z["shape"])
    occ = np.unpackbits(z["occ_packed"], axis=1)[:, :cols].astype(np.uint8)
    g = maps.GridMap(np.ascontiguousarray(occ), float(z["resolution"]),
                     tuple(float(v) for v in z["origin"]), name)
    return g, z


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    O.build()
    return O


def gpu_count():
    from pyracecarsimulator_amd import _lib
    try:
        return _lib.lib().rl_device_count()
    except Exception:
        return 0


@pytest.fixture(scope="session")
def need_gpu():
    if gpu_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box "
                    "(they never fall back to a CPU path)")
