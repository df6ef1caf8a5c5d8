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


def occupancy_grid_msg(gmap, binarise=True):
    """A nav_msgs/OccupancyGrid-shaped object for ``gmap`` the way the reference receives and
    rewrites it (/root/reference/scripts/ros_interface.py:77-86): map_server data in {-1, 0, 100}
    (unknown cells sprinkled over free space), optionally binarised to {0, 255} with ``data > 0``;
    origin pose with the yaw as a quaternion (ros_interface.py:212-216)."""
    import math
    from types import SimpleNamespace as NS
    rng = np.random.default_rng(1234)
    data = np.where(gmap.occ != 0, 100, 0).astype(np.int16)
    unknown = (rng.random(data.shape) < 0.05) & (gmap.occ == 0)
    data[unknown] = -1
    flat = data.ravel()
    if binarise:
        flat = np.where(flat > 0, 255, 0)
    yaw = float(gmap.origin[2])
    q = NS(x=0.0, y=0.0, z=math.sin(yaw / 2.0), w=math.cos(yaw / 2.0))
    info = NS(width=gmap.cols, height=gmap.rows, resolution=gmap.resolution,
              origin=NS(position=NS(x=gmap.origin[0], y=gmap.origin[1], z=0.0), orientation=q))
    return NS(info=info, data=tuple(int(v) for v in flat))
