"""pyracecarsimulator_amd — MI355X-native batched 2D lidar scan path.

Scope: ``ScanSimulator2D.scan/scanMany`` over a ``range_libc``-compatible
``PyOMap`` / ``PyRayMarching[GPU]`` / ``PyCDDTCast`` surface, implemented as hand-written
HIP kernels for gfx950 behind the C ABI in ``include/scanlib.h``.
"""
from . import maps, racecar, range_libc             # noqa: F401
from .scan_simulator import ScanSimulator2D          # noqa: F401
from .racecar_simulator import RacecarSimulator      # noqa: F401

__all__ = ["maps", "racecar", "range_libc", "ScanSimulator2D", "RacecarSimulator"]
