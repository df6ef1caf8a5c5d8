"""The two-player scan front-end (CDDT) — mirror of scripts/two_player/scan.py.

The reference's second ``ScanSimulator2D`` (scripts/two_player/scan.py:11-72) scans ONE pose per
call with ``PyCDDTCast`` and the 2-argument ``calc_range_many`` (one (x, y, theta) row per beam,
angles from ``np.arange(theta - fov/2, theta + fov/2, fov/num_rays)``, :57-70), and its caller
rebuilds map + CDDT before EVERY scan because the other car's outline is stamped into the grid
(scripts/two_player/rcs_two_player.py:110-124: ``build(map_msg, mrx, 112)`` then ``scan(*pose)``).

Same attributes and methods here.  ``build`` keeps the device objects when the new map has the
shape and world transform of the previous one and only uploads the cells (``rl_map_update``: EDT,
bit map and CDDT are rebuilt on the GPU, 0.28 ms for colombia) instead of constructing a new
``PyOMap`` / ``PyCDDTCast`` pair per scan.
"""
from __future__ import annotations

import math

import numpy as np

from . import range_libc


class ScanSimulator2D:
    def __init__(self, num_rays, fov, scan_std, batch_size=100):
        self.batch_size = batch_size
        self.num_rays = num_rays
        self.fov = fov
        self.scan_std = scan_std
        self.theta_inc = fov / num_rays
        self.twopi = math.pi * 2
        # cached vectors, as the reference keeps them (scan.py:32-35); scan() returns an alias
        self.output_vector = np.zeros(self.num_rays, dtype=np.float32)
        self.noise = np.zeros(self.num_rays, dtype=np.float32)
        self.input_vector = np.zeros((self.num_rays, 3), dtype=np.float32)
        self.omap = None
        self.scan_method = None
        self._built_for = None

    def build(self, map_msg, mrx, theta_disc):
        """``PyOMap(map_msg)`` + ``PyCDDTCast(omap, mrx, theta_disc)`` (scan.py:38-46)."""
        occ, res, org = range_libc.PyOMap._ingest(map_msg, None, None, None)
        key = (occ.shape, float(res), tuple(float(v) for v in org), float(mrx), int(theta_disc))
        if self.omap is not None and key == self._built_for:
            # same grid geometry.  The reference's caller lays the other car's outline over the SAME original map before
            # every scan (rcs_two_player.py:110-118): when the new grid is the base grid plus a few occupied cells, only
            # those cell indices cross PCIe (rl_map_stamp_cells) — otherwise the whole grid (rl_map_update)
            new = np.ascontiguousarray(occ, dtype=np.uint8) != 0
            diff = np.flatnonzero(new.reshape(-1) != self._base.reshape(-1))
            if diff.size <= self.STAMP_MAX and bool(new.reshape(-1)[diff].all()):
                self.omap.stamp_cells(diff)
            else:
                self.omap.update(occ)
                self._base = new.copy()
            return
        self.omap = range_libc.PyOMap(map_msg)
        self.scan_method = range_libc.PyCDDTCast(self.omap, mrx, theta_disc)
        self._base = np.ascontiguousarray(occ, dtype=np.uint8) != 0
        self._built_for = key

    #: build(): grids that differ from the base map by at most this many newly occupied cells are sent as cell indices
    STAMP_MAX = 4096

    def build_with_outline(self, bound_cells):
        """The same tick when the caller has the outline CELLS at hand (flat indices x * map_width + y as
        rcs_two_player.py:112-114 computes them): nothing but the indices is touched on the host."""
        self.omap.stamp_cells(bound_cells)

    def scan(self, x, y, theta):
        """One fan of ``num_rays`` beams from (x, y, theta) (scan.py:49-72)."""
        max_theta = theta + self.fov / 2.0
        min_theta = theta - self.fov / 2.0
        thetas = np.arange(min_theta, max_theta, self.theta_inc, dtype=np.float32)
        self.input_vector[:, 0] = x
        self.input_vector[:, 1] = y
        self.input_vector[:, 2] = thetas              # (raises like the reference if arange
        #                                                yields num_rays +- 1 elements)
        self.scan_method.calc_range_many(self.input_vector, self.output_vector)
        return self.output_vector
