"""The HIP detectors and grid front-end against the vectors of the independent Python witnesses
(tests/golden/witness_frontends.npz, generator tests/golden/make_golden_witness.py; tests/test_witness_cpu.py holds the
C oracle to the same vectors).  Nothing here needs the oracle or scipy at run time: the vectors travel."""
import os

import numpy as np
import pytest

from tests.detect_cases import S2B
from tests.golden.make_golden_witness import SCAN_FIELDS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wit(golden_dir):
    return np.load(os.path.join(golden_dir, "witness_frontends.npz"))


def test_detect2d_reproduces_the_witness_vectors(wit):
    from reflector_ekf_slam_amd import OdometryData
    from reflector_ekf_slam_amd.detect import LaserReflectorDetect, LaserScan, ReflectorDetectOptions
    for name in wit["d2_names"]:
        d2 = LaserReflectorDetect(ReflectorDetectOptions(), max_beams=4096, sensor_to_base_link=S2B)
        for (t, px, py, qz, qw, vx, vy, wz) in wit[f"d2_{name}_odom"]:
            d2.HandleOdometryData(OdometryData(t, (vx, vy, 0.0), (0.0, 0.0, wz), (px, py, 0.0), (qw, 0.0, 0.0, qz)))
        sc = dict(zip(SCAN_FIELDS, wit[f"d2_{name}_scalars"]))
        obs = d2.HandleLaserScan(LaserScan(ranges=wit[f"d2_{name}_ranges"], intensities=wit[f"d2_{name}_intensities"], **sc))
        want = wit[f"d2_{name}_centers"]
        assert obs.cloud_.shape == want.shape, name                              # the same reflectors ...
        if want.size:
            assert np.array_equal(obs.cloud_, want), name                        # ... bit for bit (device sin / cos = glibc's algorithm)
        ret, wret = d2.GetRangeData().returns, wit[f"d2_{name}_returns"]
        assert ret.shape == wret.shape and np.array_equal(ret, wret), name
        d2.close()


def test_detect3d_reproduces_the_witness_vectors(wit):
    from reflector_ekf_slam_amd.detect import PointCloudOptions, PointCloudReflectorDetect
    for name in wit["d3_names"]:
        d3 = PointCloudReflectorDetect(PointCloudOptions(), max_points=32768, sensor_to_base_link=tuple(wit[f"d3_{name}_s2b"]))
        o = d3.HandlePointCloud(1.5, wit[f"d3_{name}_cloud"])
        want = wit[f"d3_{name}_centers"]
        assert o.cloud_.shape == want.shape and np.abs(o.cloud_ - want).max() < 1e-6, name   # same clusters, same order
        d3.close()


def test_grid_insert_and_match_reproduce_the_witness_vectors(wit):
    from reflector_ekf_slam_amd.grid import GridFrontEnd
    from tests.grid_cases import room_grid
    res, mx, my, n, k = wit["gi_meta"]
    gf = GridFrontEnd(max_points=4096, max_cells=1024 * 1024, max_candidates=1 << 16)
    gf.SetGrid(np.zeros((int(n), int(n)), np.uint16), float(res), (float(mx), float(my)))
    for i in range(int(k)):
        gf.Insert(wit[f"gi_{i}_origin"], wit[f"gi_{i}_returns"], wit[f"gi_{i}_misses"], grow=False)
        assert np.array_equal(gf.GetGrid(), wit[f"gi_{i}_cells_after"]), i      # every cell, rays through pixel corners included
    rc, rmax, _ = room_grid()
    gf.SetGrid(rc, 0.05, rmax)
    r = gf.Match(wit["gm_init"], wit["gm_points"])
    assert list(r.best) == wit["gm_best"].tolist() and np.float32(r.score) == wit["gm_score"]
    assert np.abs(r.pose_estimate - wit["gm_pose"]).max() < 1e-12
    gf.close()
