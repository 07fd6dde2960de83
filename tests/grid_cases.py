"""Fixtures shared by the CPU (oracle) and GPU tests of the grid-mapper front-end (include/rgrid.h)."""
from __future__ import annotations

import math

import numpy as np


def room_grid(resolution=0.05, half=12.0, seed=3):
    """A probability grid of a room: an outer wall ellipse and a few pillars are occupied (low correspondence cost),
    the inside is free (high cost), a band near the border is unknown (0).  Returns (cells uint16 (ny, nx), max_xy,
    occupied points (M, 2) in the map frame)."""
    rng = np.random.default_rng(seed)
    n = int(round(2 * half / resolution))
    cells = np.full((n, n), 30000, np.uint16)                    # free: correspondence cost ~0.83 -> probability ~0.17
    cells[:8, :] = 0; cells[-8:, :] = 0; cells[:, :8] = 0; cells[:, -8:] = 0
    max_xy = (half, half)
    th = np.linspace(0, 2 * math.pi, 6000, endpoint=False)
    occ = [np.stack([9.0 * np.cos(th), 6.5 * np.sin(th)], 1)]
    for cx, cy in ((2.0, 1.5), (-3.5, 2.5), (4.0, -3.0), (-1.0, -4.0)):
        occ.append(np.stack([cx + 0.35 * np.cos(th[::10]), cy + 0.35 * np.sin(th[::10])], 1))
    occ = np.concatenate(occ)
    ix = np.rint((max_xy[1] - occ[:, 1]) / resolution - 0.5).astype(int)       # x index from y (map_limits.h:47-55)
    iy = np.rint((max_xy[0] - occ[:, 0]) / resolution - 0.5).astype(int)
    ok = (ix >= 0) & (iy >= 0) & (ix < n) & (iy < n)
    cells[iy[ok], ix[ok]] = rng.integers(1200, 2600, ok.sum()).astype(np.uint16)   # occupied: probability ~0.85
    cells[5, 100:140] = cells[5, 100:140] | 32768                              # a few values with the update marker set
    return cells, max_xy, occ


def scan_of(occ, pose, n_points=700, noise=0.01, seed=5, max_range=30.0):
    """The occupied points seen from `pose` (x, y, theta), in the tracking frame, sub-sampled and noisy."""
    rng = np.random.default_rng(seed)
    sel = rng.choice(occ.shape[0], size=min(n_points, occ.shape[0]), replace=False)
    p = occ[np.sort(sel)]
    c, s = math.cos(pose[2]), math.sin(pose[2])
    dx, dy = p[:, 0] - pose[0], p[:, 1] - pose[1]
    loc = np.stack([c * dx + s * dy, -s * dx + c * dy], 1) + rng.normal(0, noise, (p.shape[0], 2))
    keep = np.hypot(loc[:, 0], loc[:, 1]) < max_range
    return loc[keep].astype(np.float32)
