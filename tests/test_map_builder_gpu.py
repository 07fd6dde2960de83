"""mapping::MapBuilder mirrored over the C ABI (reflector_ekf_slam_amd/map_builder.py): the whole AddRangeData pipeline --
gravity alignment, voxel filters, adaptive filter, correlative match, refinement, grid creation / growth, insertion,
texture -- on the GPU against the same host logic running over the CPU oracle."""
from __future__ import annotations

import math

import numpy as np
import pytest

from tests.grid_cases import room_grid, scan_of

pytestmark = pytest.mark.gpu


def _trajectory(n):
    t = np.linspace(0.0, 1.0, n)
    return np.stack([0.4 + 2.5 * t, -0.3 + 1.2 * np.sin(2.0 * t), 0.2 + 1.4 * t], 1)


def test_add_range_data_pipeline_matches_the_oracle_composition(oracle_lib):
    from reflector_ekf_slam_amd.map_builder import MapBuilder, MapBuilderOptions, RangeData
    from tests.oracle_front_end import OracleFrontEnd
    _, _, occ = room_grid()
    gpu = MapBuilder(MapBuilderOptions(), max_points=16384, max_cells=2048 * 2048)
    ref = MapBuilder(MapBuilderOptions(), front_end=OracleFrontEnd())
    assert gpu.ToSubmapTexture() is None and gpu.AddRangeData(0.0, RangeData(np.zeros(2), np.zeros((0, 2)), np.zeros((0, 2))), (0, 0, 0)) is None
    rng = np.random.default_rng(17)
    poses = _trajectory(8)
    exact = 0
    for k, true in enumerate(poses):
        pts = scan_of(occ, true, n_points=1800, seed=300 + k).astype(np.float32)        # returns in the tracking frame
        ang = rng.uniform(-math.pi, math.pi, 40)
        misses = np.stack([6.0 * np.cos(ang), 6.0 * np.sin(ang)], 1).astype(np.float32)
        rd = RangeData(np.zeros(2, np.float32), pts, misses)
        ekf_pose = true + rng.normal(0, 1, 3) * [0.03, 0.03, 0.01]                        # what the filter would hand over
        a = gpu.AddRangeData(float(k), rd, ekf_pose)
        b = ref.AddRangeData(float(k), rd, ekf_pose)
        assert a is not None and b is not None
        pose_diff = np.abs(a.local_pose - b.local_pose).max()
        # the LM solve stops at a relative function tolerance of 1e-6: on the rare scan where it needs ~70 iterations through
        # a flat valley, rounding differences between the two reductions are amplified to ~1e-6 m (same iteration count)
        assert pose_diff < 1e-5, (k, a.local_pose - b.local_pose)
        assert np.allclose(a.range_data_in_local.returns, b.range_data_in_local.returns, atol=2e-5, rtol=0)
        if k > 0:
            assert gpu.last_summary.iterations == ref.last_summary.iterations and gpu.last_score == ref.last_score
            # stays near the truth (the map frame itself carries the first scan's pose noise)
            assert np.abs(a.local_pose[:2] - true[:2]).max() < 0.15 and abs(a.local_pose[2] - true[2]) < 0.05, (k, a.local_pose - true)
        (ga, la), (gb, lb) = gpu.grid(), ref.grid()
        assert la == lb
        if pose_diff < 1e-9:
            exact += 1
            assert np.array_equal(ga, gb), (k, int(np.count_nonzero(ga != gb)))
        else:                                                                     # a few returns fall into the neighbouring cell
            assert np.count_nonzero(ga != gb) < 0.01 * np.count_nonzero(ga), (k, int(np.count_nonzero(ga != gb)))
            ref._fe.cells = ga.copy()                                             # continue in lockstep
    assert exact >= len(poses) - 2
    assert la[0] > 100 and gpu.num_range_data == len(poses)                               # the 100 x 100 start grid has grown
    ta, tb = gpu.ToSubmapTexture(), ref.ToSubmapTexture()
    assert np.array_equal(ta["cells"], tb["cells"]) and {k: v for k, v in ta.items() if k != "cells"} == {k: v for k, v in tb.items() if k != "cells"}
    assert ta["width"] * ta["height"] * 2 == ta["cells"].size and ta["resolution"] == float(np.float32(0.05))


def test_add_range_data_c_entry_point_equals_the_python_mirror():
    """rgrid_add_range_data (one C call per scan) against reflector_ekf_slam_amd.map_builder.MapBuilder (eight calls with
    the host logic in Python): same device functions, same host arithmetic -- poses, returned clouds and grids must be
    identical bit for bit."""
    from reflector_ekf_slam_amd.grid import GridFrontEnd
    from reflector_ekf_slam_amd.map_builder import MapBuilder, MapBuilderOptions, RangeData
    _, _, occ = room_grid()
    opt = MapBuilderOptions()
    mb = MapBuilder(opt, max_points=16384, max_cells=2048 * 2048)
    fe = GridFrontEnd(max_points=16384, max_cells=2048 * 2048)
    st, _, _ = fe.AddRangeData(opt, np.zeros(2), np.zeros((0, 2), np.float32), None, (0.0, 0.0, 0.0))
    assert st == 1                                                                    # no returns: dropped
    rng = np.random.default_rng(23)
    for k, true in enumerate(_trajectory(6)):
        pts = scan_of(occ, true, n_points=1500, seed=500 + k).astype(np.float32)
        ang = rng.uniform(-math.pi, math.pi, 30)
        misses = np.stack([6.0 * np.cos(ang), 6.0 * np.sin(ang)], 1).astype(np.float32) if k % 2 == 0 else None
        ekf_pose = true + rng.normal(0, 1, 3) * [0.03, 0.03, 0.01]
        a = mb.AddRangeData(float(k), RangeData(np.zeros(2, np.float32), pts, np.zeros((0, 2), np.float32) if misses is None else misses), ekf_pose)
        st, pose, in_local = fe.AddRangeData(opt, np.zeros(2, np.float32), pts, misses, ekf_pose)
        assert st == 0 and np.array_equal(pose, a.local_pose), (k, pose - a.local_pose)
        assert np.array_equal(in_local, a.range_data_in_local.returns)
        ga, la = mb.grid()
        assert fe.GetLimits() == la and np.array_equal(fe.GetGrid(), ga)
    fe.close()
