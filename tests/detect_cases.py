"""Scan / cloud fixtures shared by the CPU (oracle) and GPU (HIP path) detector tests."""
from __future__ import annotations

import math
from types import SimpleNamespace as NS

import numpy as np

from reflector_ekf_slam_amd import synth

S2B = (0.13686, 0.0, 0.0)          # launch/slam.launch:27


def world_scan(seed=1, pose=(16.0, 17.7, 0.6), stamp=10.0, n_beams=3600):
    cfg = synth.C2
    rng = np.random.Generator(np.random.PCG64(seed))
    lms = synth.make_world(cfg, rng)
    return NS(**synth.make_laser_scan(lms, pose, stamp, rng, n_beams=n_beams)), lms


def plate_scan(n_beams=720, plates=(), stamp=5.0, scan_time=0.1, base_range=20.0, base_int=50.0):
    """Hand-built scan: plates = [(first_beam, n, range, intensity)], everything else far and dim.
    Adjacent beams of a plate at range r are r*inc apart: n beams span (n-1)*r*inc metres."""
    inc = np.float32(2.0 * math.pi / n_beams)
    r = np.full(n_beams, base_range, np.float32)
    it = np.full(n_beams, base_int, np.float32)
    for first, n, rng_, inten in plates:
        for b in range(first, first + n):
            r[b % n_beams] = rng_
            it[b % n_beams] = inten
    amin = np.float32(-math.pi)
    return NS(stamp=stamp, angle_min=float(amin), angle_max=float(amin + inc * (n_beams - 1)),
              angle_increment=float(inc), scan_time=scan_time, range_min=0.05, range_max=30.0,
              ranges=r, intensities=it)


def beams_for_width(width, rng_, n_beams):
    """How many beams a plate of `width` metres at range rng_ covers (end-to-end distance ~ width)."""
    inc = 2.0 * math.pi / n_beams
    return int(round(width / (rng_ * inc))) + 1


def odom_stream(t0, t1, hz=50.0, v=1.0, w=0.3):
    """Straight-ish drive: list of (t, px, py, qz, qw, vx, vy, wz)."""
    out = []
    x = y = th = 0.0
    t = t0
    dt = 1.0 / hz
    while t <= t1 + 1e-9:
        out.append((t, x, y, math.sin(th / 2), math.cos(th / 2), v, 0.0, w))
        x += v * dt * math.cos(th)
        y += v * dt * math.sin(th)
        th += w * dt
        t += dt
    return out
