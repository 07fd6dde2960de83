"""Generates tests/golden/witness_frontends.npz: input / output vectors of the detectors and the grid front-end whose
EXPECTED values come from the independent Python witnesses under tests/witness/ (scipy kd-tree / numpy / exact rational
arithmetic), NOT from the C oracle.  The CPU suite checks oracle == these vectors, the GPU suite checks the HIP path against
them.  The reference ships no fixtures of its own (parity with it stays unpinned): this is the strongest pin available.

    python -m tests.golden.make_golden_witness
"""
from __future__ import annotations

import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from reflector_ekf_slam_amd import synth                                           # noqa: E402
from tests.detect_cases import S2B, beams_for_width, odom_stream, plate_scan, world_scan   # noqa: E402
from tests.grid_cases import room_grid, scan_of                                    # noqa: E402
from tests.witness.detect2d_witness import Extrapolator, detect2d_witness          # noqa: E402
from tests.witness.detect3d_witness import detect3d_witness                        # noqa: E402
from tests.witness.grid_witness import insert_witness, match_witness               # noqa: E402

SCAN_FIELDS = ("stamp", "angle_min", "angle_max", "angle_increment", "scan_time", "range_min", "range_max")


def cases_2d():
    n = 720
    nb = beams_for_width(0.18, 3.0, n)
    out = []
    s, _ = world_scan(seed=1)
    out.append(("world_still", s, []))
    s, _ = world_scan(seed=2)
    out.append(("world_moving", s, odom_stream(9.8, 10.05)))
    out.append(("two_plates", plate_scan(n, [(100, nb, 3.0, 200.0), (300, nb, 5.0, 220.0)]), []))
    out.append(("wrap_merge", plate_scan(n, [(0, nb, 3.0, 200.0), (n - 3, 3, 3.0, 200.0)]), odom_stream(4.85, 5.05)))
    out.append(("gap_bridge", plate_scan(n, [(100, 3, 3.0, 200.0), (105, 3, 3.05, 200.0)]), []))
    out.append(("nothing_bright", plate_scan(n, []), []))
    return out


def main():
    d = {}
    names2 = []
    for name, scan, odom in cases_2d():
        ex = Extrapolator()
        for rec in odom:
            ex.add(rec)
        c, members, ret = detect2d_witness(scan, ex, s2b=S2B)
        names2.append(name)
        d[f"d2_{name}_scalars"] = np.array([getattr(scan, k) for k in SCAN_FIELDS], np.float64)
        d[f"d2_{name}_ranges"] = np.asarray(scan.ranges, np.float32)
        d[f"d2_{name}_intensities"] = np.asarray(scan.intensities, np.float32)
        d[f"d2_{name}_odom"] = np.array(odom, np.float64).reshape(-1, 8)
        d[f"d2_{name}_centers"] = c
        d[f"d2_{name}_returns"] = ret
        d[f"d2_{name}_members"] = np.concatenate(members) if members else np.zeros(0, np.int32)
        d[f"d2_{name}_member_off"] = np.cumsum([0] + [m.size for m in members]).astype(np.int32)
    d["d2_names"] = np.array(names2)
    rng = np.random.Generator(np.random.PCG64(33))
    lms = synth.make_world(synth.C4, rng)
    names3 = []
    for name, pose, kw, s2b in (("c4_lidar", (10.0, 50.0, -0.4), synth.C4_LIDAR, (0.0, 0.0, 0.0)),
                                ("wide_lidar_offset_mount", (34.4, 34.0, 1.15), dict(rings=16, n_az=600, max_range=15.0), (0.3, -0.2, 0.7))):
        cloud = synth.make_point_cloud(lms, pose, rng, **kw)
        c, m1, m2 = detect3d_witness(cloud, sensor_to_base_link=s2b)
        names3.append(name)
        d[f"d3_{name}_cloud"] = cloud
        d[f"d3_{name}_s2b"] = np.array(s2b)
        d[f"d3_{name}_centers"] = c
        d[f"d3_{name}_counts"] = np.array([m1, m2], np.int32)
    d["d3_names"] = np.array(names3)
    # ---- grid: three insertions into an initially unknown grid (rays between cell centres included), one match
    g = np.random.default_rng(4)
    res, n, max_xy = 0.05, 160, (4.0, 4.0)
    cells = np.zeros((n, n), np.uint16)
    for k in range(3):
        origin = np.array([0.0125 + 0.05 * g.integers(-10, 10), 0.0125 + 0.05 * g.integers(-10, 10)], np.float32)
        ret = g.uniform(-3.6, 3.6, (120, 2)).astype(np.float32)
        ret[:30] = (0.025 + 0.05 * g.integers(-60, 60, (30, 2))).astype(np.float32)
        mis = g.uniform(-3.6, 3.6, (15, 2)).astype(np.float32)
        new = insert_witness(cells, res, max_xy, origin, ret, mis)
        d[f"gi_{k}_origin"], d[f"gi_{k}_returns"], d[f"gi_{k}_misses"], d[f"gi_{k}_cells_after"] = origin, ret, mis, new
        cells = new
    d["gi_meta"] = np.array([res, max_xy[0], max_xy[1], n, 3])
    rc, rmax, occ = room_grid()
    pose = (0.3, -0.2, 0.1)
    pts = scan_of(occ, pose, n_points=300)
    init = (pose[0] + 0.07, pose[1] - 0.05, pose[2] + 0.03)
    score, est, best = match_witness(init, pts, rc, 0.05, rmax)
    d["gm_points"], d["gm_init"], d["gm_score"], d["gm_pose"], d["gm_best"] = pts, np.array(init), np.float32(score), np.array(est), np.array(best, np.int32)
    path = os.path.join(ROOT, "tests", "golden", "witness_frontends.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
