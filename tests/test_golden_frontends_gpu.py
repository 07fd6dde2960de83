"""HIP detectors and grid front-end against the committed fixtures (tests/golden/frontends.npz): the vectors travel to
the GPU box, nothing here needs the oracle at run time."""
import os

import numpy as np
import pytest

from tests.detect_cases import S2B

pytestmark = pytest.mark.gpu


def _load(golden_dir):
    return np.load(os.path.join(golden_dir, "frontends.npz"))


def test_detectors_reproduce_the_fixtures(golden_dir):
    from reflector_ekf_slam_amd import OdometryData
    from reflector_ekf_slam_amd.detect import (LaserReflectorDetect, LaserScan, PointCloudOptions, PointCloudReflectorDetect,
                                               ReflectorDetectOptions)
    g = _load(golden_dir)
    m = g["d2_meta"]
    d2 = LaserReflectorDetect(ReflectorDetectOptions(), max_beams=4096, sensor_to_base_link=S2B)
    for (t, px, py, qz, qw, vx, vy, wz) in g["d2_odom"]:
        d2.HandleOdometryData(OdometryData(t, (vx, vy, 0.0), (0.0, 0.0, wz), (px, py, 0.0), (qw, 0.0, 0.0, qz)))
    obs = d2.HandleLaserScan(LaserScan(float(m[0]), float(m[1]), float(m[2]), float(m[3]), float(m[4]), float(m[5]), float(m[6]),
                                       g["d2_ranges"], g["d2_intens"]))
    assert obs.cloud_.shape == g["d2_centres"].shape and np.array_equal(obs.cloud_, g["d2_centres"])
    ret = d2.GetRangeData().returns
    assert ret.shape == g["d2_returns"].shape and np.array_equal(ret, g["d2_returns"])
    d3 = PointCloudReflectorDetect(PointCloudOptions(), max_points=16384, sensor_to_base_link=(0.2, -0.1, 0.3))
    o3 = d3.HandlePointCloud(2.5, g["d3_cloud"])
    assert o3.cloud_.shape == g["d3_centres"].shape and np.abs(o3.cloud_ - g["d3_centres"]).max() < 1e-5


def test_grid_front_end_reproduces_the_fixtures(golden_dir):
    from reflector_ekf_slam_amd.grid import AdaptiveVoxelFilterOptions, GridFrontEnd
    g = _load(golden_dir)
    res, mx, my = (float(v) for v in g["g_meta"])
    gf = GridFrontEnd(max_points=4096, max_cells=1024 * 1024, max_candidates=1 << 16)
    pts, init = g["g_pts"], g["g_init"]
    vf = gf.VoxelFilter(pts, 0.05)
    assert np.array_equal(vf, g["g_voxel"])
    assert np.array_equal(gf.AdaptiveVoxelFilter(pts, AdaptiveVoxelFilterOptions(0.5, 120, 50.0)), g["g_adaptive"])
    gf.SetGrid(g["g_cells"], res, (mx, my))
    r = gf.Match(init, vf)
    assert list(r.best) == g["g_best"].tolist() and list(r.info) == g["g_info"].tolist()
    assert abs(r.score - g["g_match"][0]) <= 1e-7 * g["g_match"][0] and np.abs(r.pose_estimate - g["g_match"][1:]).max() < 1e-12
    f = gf.RefineMatch(init[:2], r.pose_estimate, vf)
    assert [f.iterations, f.termination] == g["g_refine"][4:].astype(int).tolist()
    assert np.abs(f.pose_estimate - g["g_refine"][:3]).max() < 1e-8 and abs(f.final_cost - g["g_refine"][3]) < 1e-8 * g["g_refine"][3]
    tex, box, sm = gf.DrawTexture()
    assert np.array_equal(tex, g["g_tex"]) and list(box) == g["g_box"].tolist() and list(sm) == g["g_slice"].tolist()
    gf.GrowAsNeeded(np.zeros(2, np.float32), np.array([[7.0, 1.0]], np.float32))
    lim = gf.GetLimits()
    assert [lim[1], lim[0]] == g["g_grown_shape"].tolist() and [lim[3], lim[4]] == g["g_grown_max"].tolist()
    grown = gf.GetGrid()
    off = g["g_grown_off"]
    assert np.array_equal(grown[off[1]:off[1] + 200, off[0]:off[0] + 200], g["g_cells"]) and np.count_nonzero(grown) == np.count_nonzero(g["g_cells"])
    gf.close()
