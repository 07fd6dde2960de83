"""Generates tests/golden/frontends.npz: input / output vectors of the detectors and the grid front-end.

Unlike the EKF fixtures (independent numpy restatement) these come from the C oracle itself: the reference cannot be
built or imported here and ships no fixtures, so the file does not add an independent witness -- it pins the oracle
(any later change to it shows up in tests/test_golden_frontends_cpu.py) and gives the GPU tests vectors that travel.
Run from the repo root:   python tests/golden/make_golden_frontends.py
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from reflector_ekf_slam_amd import synth  # noqa: E402
from tests.detect_cases import S2B, odom_stream, world_scan  # noqa: E402
from tests.grid_cases import room_grid, scan_of  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def build():
    out = {}
    # ---- 2D detector: a 1440-beam world scan with odometry (de-skew)
    sc, _ = world_scan(seed=11, pose=(14.0, 20.0, -0.7), n_beams=1440)
    odom = np.array(odom_stream(sc.stamp - 0.3, sc.stamp + 0.05, v=1.2, w=0.4))
    o2 = ob.OracleDetect2D(sensor_to_base_link=S2B)
    for row in odom:
        o2.handle_odometry(*row)
    t, centres = o2.handle_scan(sc)
    out.update(d2_ranges=sc.ranges, d2_intens=sc.intensities, d2_odom=odom, d2_centres=centres, d2_returns=o2.returns(),
               d2_meta=np.array([sc.stamp, sc.angle_min, sc.angle_max, sc.angle_increment, sc.scan_time, sc.range_min, sc.range_max]))
    # ---- 3D detector: an 8-ring sweep of the C4 world
    rng = np.random.Generator(np.random.PCG64(5))
    lms = synth.make_world(synth.C4, rng)
    cloud = synth.make_point_cloud(lms, (30.0, 28.0, 0.9), rng, rings=8, n_az=900).astype(np.float32)
    c3, m1, m2 = ob.oracle_detect3d(cloud, sensor_to_base_link=(0.2, -0.1, 0.3))
    out.update(d3_cloud=cloud, d3_centres=c3, d3_counts=np.array([m1, m2]))
    # ---- grid front-end on a 200 x 200 map built by the inserter
    _, _, occ = room_grid()
    res = 0.05
    max_xy = (5.0, 5.0)
    cells = np.zeros((200, 200), np.uint16)
    for k, pose in enumerate(((0.0, 0.0, 0.0), (0.6, -0.4, 0.5))):
        loc = scan_of(occ, pose, n_points=700, seed=80 + k, max_range=4.5)
        c, s = math.cos(pose[2]), math.sin(pose[2])
        world = np.stack([pose[0] + c * loc[:, 0] - s * loc[:, 1], pose[1] + s * loc[:, 0] + c * loc[:, 1]], 1).astype(np.float32)
        cells = ob.oracle_insert(cells, res, max_xy, np.array(pose[:2], np.float32), world)
    true = np.array([0.3, -0.2, 0.25])
    pts = scan_of(occ, true, n_points=2400, seed=91, max_range=4.5).astype(np.float32)
    vf = ob.oracle_voxel_filter(pts, 0.05)
    av = ob.oracle_adaptive_voxel_filter(pts, 0.5, 120, 50.0)
    init = true + [0.06, -0.05, 0.03]
    score, pose, best, info = ob.oracle_match(init, vf, cells, res, max_xy)
    rpose, rs = ob.oracle_refine_match(init[:2], pose, vf, cells, res, max_xy)
    tex, box, sm = ob.oracle_draw_texture(cells, res, max_xy)
    grown, gmax, goff = ob.oracle_grow(cells, res, max_xy, np.zeros(2, np.float32), np.array([[7.0, 1.0]], np.float32))
    out.update(g_cells=cells, g_pts=pts, g_voxel=vf, g_adaptive=av, g_init=init, g_match=np.array([score, *pose]), g_best=np.array(best),
               g_info=np.array(info), g_refine=np.array([*rpose, rs["final_cost"], rs["iterations"], rs["termination"]]),
               g_tex=tex, g_box=np.array(box), g_slice=np.array(sm), g_grown_shape=np.array(grown.shape), g_grown_max=np.array(gmax),
               g_grown_off=np.array(goff), g_meta=np.array([res, max_xy[0], max_xy[1]]))
    return out


if __name__ == "__main__":
    d = build()
    np.savez_compressed(os.path.join(HERE, "frontends.npz"), **d)
    print({k: getattr(v, "shape", None) for k, v in d.items()})
