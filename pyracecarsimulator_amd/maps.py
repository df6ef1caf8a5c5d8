"""Map ingestion for the lidar scan path (host side, NumPy only).

Replaces the ROS pieces in front of ``range_libc.PyOMap``:

* ``map_server`` PGM+YAML loading (formats: /root/reference/maps/colombia/map.pgm:1-4,
  maps/colombia/map.yaml:1-7) with map_server's occupancy thresholds and its
  vertical flip (image row 0 is the TOP of the map, grid row 0 the bottom);
* the reference's binarisation ``data > 0 -> 255 else 0`` (unknown -1 => free)
  at scripts/ros_interface.py:80-86 and scripts/mcts_driver.py:106-112;
* range_libc's own test ``data > 10`` in PyOMap.

Also holds the deterministic synthetic inputs BASELINE.json's configs name
(``make_maze``, ``sample_free_poses``), generated with ``numpy.random.default_rng``
so the GPU box can rebuild them from a seed.
"""
from __future__ import annotations

import os
import re
from dataclasses import dataclass

import numpy as np

__all__ = [
    "GridMap", "read_pgm", "read_map_yaml", "load_map_server_map", "occupancy_from_image",
    "binarise_reference", "make_maze", "make_room", "sample_free_poses", "load_colombia",
]


@dataclass
class GridMap:
    """Binary occupancy grid plus the map_server world transform.

    ``occ[r, c]`` (uint8, 0/1): r = row = world y, c = col = world x, row 0 = min y.
    """
    occ: np.ndarray
    resolution: float
    origin: tuple  # (x, y, yaw)
    name: str = "map"

    @property
    def rows(self) -> int:
        return int(self.occ.shape[0])

    @property
    def cols(self) -> int:
        return int(self.occ.shape[1])


def read_pgm(path: str) -> np.ndarray:
    """Read a P2 (ASCII) or P5 (binary) PGM into a (H, W) uint8/uint16 array."""
    with open(path, "rb") as f:
        raw = f.read()
    # header: magic, width, height, maxval; '#' comments allowed between tokens
    tokens = []
    pos = 0
    while len(tokens) < 4:
        m = re.compile(rb"\s*(#[^\n]*\n|\S+)").match(raw, pos)
        if m is None:
            raise ValueError("truncated PGM header: %s" % path)
        pos = m.end()
        if not m.group(1).startswith(b"#"):
            tokens.append(m.group(1))
    magic, w, h, maxval = tokens[0], int(tokens[1]), int(tokens[2]), int(tokens[3])
    dtype = np.uint8 if maxval < 256 else np.dtype(">u2")
    if magic == b"P5":
        pos += 1  # single whitespace byte after maxval
        img = np.frombuffer(raw, dtype=dtype, count=w * h, offset=pos).reshape(h, w)
    elif magic == b"P2":
        body = re.sub(rb"#[^\n]*", b"", raw[pos:])
        img = np.array(body.split(), dtype=np.int64)
        if img.size != w * h:
            raise ValueError("PGM %s: expected %d samples, got %d" % (path, w * h, img.size))
        img = img.reshape(h, w).astype(np.uint8 if maxval < 256 else np.uint16)
    else:
        raise ValueError("not a PGM (P2/P5): %s" % path)
    return np.ascontiguousarray(img)


def read_map_yaml(path: str) -> dict:
    """Parse the flat map_server YAML (image, resolution, origin, negate, thresholds)."""
    import yaml
    with open(path) as f:
        y = yaml.safe_load(f)
    return {
        "image": y["image"],
        "resolution": float(y["resolution"]),
        "origin": tuple(float(v) for v in y["origin"]),
        "negate": int(y.get("negate", 0)),
        "occupied_thresh": float(y.get("occupied_thresh", 0.65)),
        "free_thresh": float(y.get("free_thresh", 0.196)),
    }


def occupancy_from_image(img: np.ndarray, negate: int = 0, occupied_thresh: float = 0.65,
                         free_thresh: float = 0.196) -> np.ndarray:
    """map_server trinary interpretation -> OccupancyGrid ``data`` as int8 (H, W).

    ``occ = (255 - p)/255`` (``p/255`` if negate); > occupied_thresh -> 100,
    < free_thresh -> 0, otherwise -1; the image is flipped vertically so row 0 is
    the bottom of the map.
    """
    p = img.astype(np.float64) / float(255 if img.dtype == np.uint8 else 65535)
    occ = p if negate else 1.0 - p
    data = np.full(img.shape, -1, dtype=np.int8)
    data[occ > occupied_thresh] = 100
    data[occ < free_thresh] = 0
    return np.ascontiguousarray(data[::-1])


def binarise_reference(data: np.ndarray) -> np.ndarray:
    """scripts/ros_interface.py:80-86 then PyOMap's ``> 10``: 1 where data > 0."""
    ref = np.where(np.asarray(data) > 0, 255, 0)          # ros_interface.py:82-85
    return np.ascontiguousarray((ref > 10).astype(np.uint8))  # PyOMap (row a6)


def load_map_server_map(yaml_path: str) -> GridMap:
    meta = read_map_yaml(yaml_path)
    img = read_pgm(os.path.join(os.path.dirname(yaml_path), meta["image"]))
    data = occupancy_from_image(img, meta["negate"], meta["occupied_thresh"], meta["free_thresh"])
    return GridMap(binarise_reference(data), meta["resolution"], meta["origin"],
                   os.path.basename(os.path.dirname(yaml_path)) or "map")


_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def load_colombia() -> GridMap:
    """The one map the reference mount still holds (maps/colombia/map.pgm, 435x350,
    0.05 m/px), shipped as data/colombia_map.npz (image bytes + YAML values; made by
    tests/golden/make_fixtures.py)."""
    z = np.load(os.path.join(_DATA, "colombia_map.npz"))
    data = occupancy_from_image(z["image"], int(z["negate"]), float(z["occupied_thresh"]),
                                float(z["free_thresh"]))
    return GridMap(binarise_reference(data), float(z["resolution"]),
                   tuple(float(v) for v in z["origin"]), "colombia")


def make_maze(n: int, cell: int = 40, wall: int = 3, p: float = 0.45, seed: int = 1,
              resolution: float = 0.05, origin=None) -> GridMap:
    """Seeded synthetic maze: an n x n grid with a solid border and
    ``wall``-thick wall segments on a ``cell``-pitch lattice, each present with
    probability ``p``.  ~7 % occupied at the default parameters."""
    rng = np.random.default_rng(seed)
    occ = np.zeros((n, n), dtype=np.uint8)
    occ[:wall, :] = 1
    occ[-wall:, :] = 1
    occ[:, :wall] = 1
    occ[:, -wall:] = 1
    k = (n + cell - 1) // cell
    horiz = rng.random((k, k)) < p
    vert = rng.random((k, k)) < p
    for i in range(1, k):
        for j in range(k):
            if horiz[i, j]:
                occ[i * cell:i * cell + wall, j * cell:min(n, (j + 1) * cell + wall)] = 1
            if vert[i, j]:
                occ[j * cell:min(n, (j + 1) * cell + wall), i * cell:i * cell + wall] = 1
    if origin is None:
        origin = (-0.5 * n * resolution, -0.5 * n * resolution, 0.0)
    return GridMap(occ, float(resolution), tuple(float(v) for v in origin), "maze%d_s%d" % (n, seed))


def make_room(n: int, wall: int = 1, resolution: float = 0.05, origin=(0.0, 0.0, 0.0)) -> GridMap:
    """Empty room with ``wall``-thick border walls (known-answer maps)."""
    occ = np.zeros((n, n), dtype=np.uint8)
    occ[:wall, :] = 1
    occ[-wall:, :] = 1
    occ[:, :wall] = 1
    occ[:, -wall:] = 1
    return GridMap(occ, float(resolution), tuple(float(v) for v in origin), "room%d" % n)


def sample_free_poses(gmap: GridMap, n_poses: int, seed: int, min_clear_px: float = 2.0,
                      dt: np.ndarray | None = None) -> np.ndarray:
    """Seeded world poses float32 (n,3): cells uniform over free cells whose distance to
    the nearest obstacle is >= ``min_clear_px`` (needs ``dt`` in cells; without it any
    free cell qualifies), sub-cell offset U[0,1)^2, heading U(-pi,pi)."""
    rng = np.random.default_rng(seed)
    free = gmap.occ == 0
    if dt is not None:
        free &= np.asarray(dt).reshape(gmap.occ.shape) >= min_clear_px
    rr, cc = np.nonzero(free)
    if rr.size == 0:
        raise ValueError("map has no free cell")
    pick = rng.integers(0, rr.size, size=n_poses)
    off = rng.random((n_poses, 2))
    th = rng.uniform(-np.pi, np.pi, size=n_poses)
    ox, oy, yaw = gmap.origin
    gx = cc[pick] + off[:, 0]
    gy = rr[pick] + off[:, 1]
    # grid -> world: rotate by +yaw, scale, translate (inverse of the world->grid map)
    c, s = np.cos(yaw), np.sin(yaw)
    xw = ox + (c * gx - s * gy) * gmap.resolution
    yw = oy + (s * gx + c * gy) * gmap.resolution
    return np.ascontiguousarray(np.stack([xw, yw, th + yaw], axis=1).astype(np.float32))
