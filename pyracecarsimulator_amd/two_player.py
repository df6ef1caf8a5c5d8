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
            # same grid geometry: the device objects stay, the grid is re-uploaded and every table rebuilt in place.
            # (Finding the outline by comparing the new grid with the previous one on the host costs more than the upload
            #  it would save — 183 vs 171 us per tick on colombia, profiles/r06/cddt_latency.txt —: a caller that HAS the
            #  outline cells uses build_with_outline.)
            self.omap.update(occ)
            return
        self.omap = range_libc.PyOMap(map_msg)
        self.scan_method = range_libc.PyCDDTCast(self.omap, mrx, theta_disc)
        self._built_for = key

    def build_with_outline(self, bound_cells):
        """The tick of rcs_two_player.py:105-121 without the grid crossing PCIe: ``bound_cells`` are the flat indices
        ``x * map_width + y`` that :112-114 computes from ``Car::getBound``'s points; they are laid over the map ``build``
        last uploaded (rl_map_stamp_cells: a stamp replaces the previous one, like ``ego_map[:] = org_map``) and every
        table is rebuilt on the device."""
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
