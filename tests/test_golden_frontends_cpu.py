"""The committed front-end fixtures (tests/golden/frontends.npz, made by tests/golden/make_golden_frontends.py) against
the oracle as it is now: any drift of the oracle shows up here."""
import os

import numpy as np

from tests.detect_cases import S2B


def _load(golden_dir):
    return np.load(os.path.join(golden_dir, "frontends.npz"))


def test_detector_fixtures(oracle_lib, golden_dir):
    from types import SimpleNamespace as NS
    from oracle import binding as ob
    g = _load(golden_dir)
    m = g["d2_meta"]
    sc = NS(stamp=float(m[0]), angle_min=float(m[1]), angle_max=float(m[2]), angle_increment=float(m[3]), scan_time=float(m[4]),
            range_min=float(m[5]), range_max=float(m[6]), ranges=g["d2_ranges"], intensities=g["d2_intens"])
    o2 = ob.OracleDetect2D(sensor_to_base_link=S2B)
    for row in g["d2_odom"]:
        o2.handle_odometry(*row)
    _, centres = o2.handle_scan(sc)
    assert np.array_equal(centres, g["d2_centres"]) and np.array_equal(o2.returns(), g["d2_returns"])
    c3, m1, m2 = ob.oracle_detect3d(g["d3_cloud"], sensor_to_base_link=(0.2, -0.1, 0.3))
    assert np.array_equal(c3, g["d3_centres"]) and [m1, m2] == g["d3_counts"].tolist()


def test_grid_fixtures(oracle_lib, golden_dir):
    from oracle import binding as ob
    g = _load(golden_dir)
    res, mx, my = g["g_meta"]
    cells, pts, init = g["g_cells"], g["g_pts"], g["g_init"]
    vf = ob.oracle_voxel_filter(pts, 0.05)
    assert np.array_equal(vf, g["g_voxel"])
    assert np.array_equal(ob.oracle_adaptive_voxel_filter(pts, 0.5, 120, 50.0), g["g_adaptive"])
    score, pose, best, info = ob.oracle_match(init, vf, cells, res, (mx, my))
    assert np.array_equal(np.array([score, *pose]), g["g_match"]) and list(best) == g["g_best"].tolist() and list(info) == g["g_info"].tolist()
    rpose, rs = ob.oracle_refine_match(init[:2], pose, vf, cells, res, (mx, my))
    assert np.allclose(np.array([*rpose, rs["final_cost"]]), g["g_refine"][:4], rtol=0, atol=1e-12)
    assert [rs["iterations"], rs["termination"]] == g["g_refine"][4:].astype(int).tolist()
    tex, box, sm = ob.oracle_draw_texture(cells, res, (mx, my))
    assert np.array_equal(tex, g["g_tex"]) and list(box) == g["g_box"].tolist() and list(sm) == g["g_slice"].tolist()
    grown, gmax, goff = ob.oracle_grow(cells, res, (mx, my), np.zeros(2, np.float32), np.array([[7.0, 1.0]], np.float32))
    assert list(grown.shape) == g["g_grown_shape"].tolist() and list(gmax) == g["g_grown_max"].tolist() and list(goff) == g["g_grown_off"].tolist()
