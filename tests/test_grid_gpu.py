"""GPU parity of the grid front-end (include/rgrid.h, csrc/rgrid.hip) against the CPU oracle, through the C ABI."""
from __future__ import annotations

import math

import numpy as np
import pytest

from tests.grid_cases import room_grid, scan_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gf():
    from reflector_ekf_slam_amd.grid import GridFrontEnd
    g = GridFrontEnd(max_points=16384, max_cells=1024 * 1024, max_candidates=1 << 18)
    yield g
    g.close()


def test_voxel_filter_matches_oracle_bit_for_bit(gf, oracle_lib):
    from oracle.binding import oracle_voxel_filter
    rng = np.random.default_rng(11)
    for n, span in ((1, 1.0), (257, 2.0), (3600, 8.0), (12000, 20.0)):
        pts = rng.uniform(-span, span, (n, 2)).astype(np.float32)
        pts[::7] = pts[::7].round(1)                                   # exact voxel-boundary values
        for res in (0.025, 0.11, 0.9):
            assert np.array_equal(gf.VoxelFilter(pts, res), oracle_voxel_filter(pts, res))
    assert gf.VoxelFilter(np.zeros((0, 2), np.float32), 0.1).shape == (0, 2)


def test_adaptive_voxel_filter_matches_oracle(gf, oracle_lib):
    from oracle.binding import oracle_adaptive_voxel_filter
    from reflector_ekf_slam_amd.grid import AdaptiveVoxelFilterOptions
    rng = np.random.default_rng(12)
    pts = rng.uniform(-9, 9, (5000, 2)).astype(np.float32)
    dense = np.tile(np.array([[1.0, 1.0]], np.float32), (600, 1)) + rng.normal(0, 1e-4, (600, 2)).astype(np.float32)
    for cloud, opt in ((pts[:300], AdaptiveVoxelFilterOptions()), (pts[:300], AdaptiveVoxelFilterOptions(max_range=5.0)),
                       (pts, AdaptiveVoxelFilterOptions(max_length=0.3)), (pts, AdaptiveVoxelFilterOptions(max_length=2.0)),
                       (pts, AdaptiveVoxelFilterOptions(max_length=2.0, min_num_points=1500, max_range=8.0)),
                       (dense, AdaptiveVoxelFilterOptions(min_num_points=550))):
        got = gf.AdaptiveVoxelFilter(cloud, opt)
        exp = oracle_adaptive_voxel_filter(cloud, opt.max_length, opt.min_num_points, opt.max_range)
        assert np.array_equal(got, exp)


def test_adaptive_voxel_filter_search_paths(gf, oracle_lib):
    """The speculative search (the halving ladder in one batch, the bisection tree several levels at a time -- 5, 3 or 2
    depending on the cloud size) must end on the voxel size the reference's sequential loop ends on: random clouds of
    every size class, thresholds that stop the search at every stage (first size, some rung of the ladder, nothing dense
    enough, deep bisection), wall-like and blob-like point sets."""
    from oracle.binding import oracle_adaptive_voxel_filter
    from reflector_ekf_slam_amd.grid import AdaptiveVoxelFilterOptions
    rng = np.random.default_rng(33)
    seen = set()
    for trial, n in enumerate((600, 800, 1500, 2600, 4000, 4097, 6000, 8192, 8193, 12000, 16000, 900, 3000, 5000)):
        if trial % 3 == 0:                                                      # walls of a room
            t = rng.uniform(0, 4, n)
            side = np.floor(t).astype(int)
            u = (t - side) * 16 - 8
            pts = np.stack([np.where(side % 2 == 0, u, np.where(side == 1, 8.0, -8.0)), np.where(side % 2 == 1, u, np.where(side == 0, -8.0, 8.0))], 1)
            pts += rng.normal(0, 0.01, pts.shape)
        elif trial % 3 == 1:
            pts = rng.uniform(-12, 12, (n, 2))
        else:
            pts = rng.normal(0, 2.5, (n, 2))
        pts = pts.astype(np.float32)
        for frac, max_length in ((0.9, 0.9), (0.5, 0.9), (0.2, 2.0), (0.05, 0.5), (0.999, 0.3), (0.7, 5.0)):
            opt = AdaptiveVoxelFilterOptions(max_length=max_length, min_num_points=max(2.0, round(frac * n)), max_range=50.0)
            got = gf.AdaptiveVoxelFilter(pts, opt)
            exp = oracle_adaptive_voxel_filter(pts, opt.max_length, opt.min_num_points, opt.max_range)
            assert np.array_equal(got, exp), (n, frac, max_length, got.shape, exp.shape)
            seen.add((got.shape[0] >= opt.min_num_points, got.shape[0] == n))
    assert len(seen) >= 3                                                        # dense enough / never dense enough / untouched


@pytest.mark.parametrize("true,dinit,npts", [((0.8, -0.6, 0.35), (0.10, -0.15, 4.0), 700), ((-2.0, 1.2, -1.9), (-0.12, 0.08, -7.0), 500),
                                             ((0.0, 0.0, 3.0), (0.0, 0.0, 0.0), 300), ((3.1, 2.2, 0.9), (0.19, 0.19, 14.0), 900)])
def test_match_identical_candidate_and_score(gf, oracle_lib, true, dinit, npts):
    """Same best candidate (scan, x offset, y offset) as the oracle, same float32 score (the exp() of the penalty is
    evaluated in FP64 on the device and on the host: tolerance one float32 ulp), same pose estimate."""
    from oracle.binding import oracle_match
    cells, max_xy, occ = room_grid()
    gf.SetGrid(cells, 0.05, max_xy)
    true = np.array(true)
    pts = scan_of(occ, true, n_points=npts, seed=int(10 * abs(true[0]) + npts))
    init = true + np.array([dinit[0], dinit[1], math.radians(dinit[2])])
    r = gf.Match(init, pts)
    score, pose, best, info = oracle_match(init, pts, cells, 0.05, max_xy)
    assert r.info == info and r.best == best
    assert abs(r.score - score) <= 1.2e-7 * score
    assert np.abs(r.pose_estimate - pose).max() < 1e-12
    assert np.abs(pose[:2] - true[:2]).max() <= 0.1


def test_match_options_windows_and_errors(gf, oracle_lib):
    from oracle.binding import oracle_match
    from reflector_ekf_slam_amd.grid import RealTimeCorrelativeScanMatcherOptions, RgridError
    cells, max_xy, occ = room_grid(resolution=0.1, half=10.0)
    gf.SetGrid(cells, 0.1, max_xy)
    true = np.array([1.0, 1.0, 0.2])
    pts = scan_of(occ, true, n_points=400)
    o = RealTimeCorrelativeScanMatcherOptions(linear_search_window=0.35, angular_search_window=math.radians(6.0),
                                              translation_delta_cost_weight=2.0, rotation_delta_cost_weight=5.0)
    r = gf.Match(true + [0.2, -0.1, 0.03], pts, o)
    score, pose, best, info = oracle_match(true + [0.2, -0.1, 0.03], pts, cells, 0.1, max_xy, 0.35, math.radians(6.0), 2.0, 5.0)
    assert r.info == info and r.best == best and abs(r.score - score) <= 1.2e-7 * score
    assert info[1] == 4                                                            # ceil(0.35 / 0.1)
    with pytest.raises(RgridError) as e:
        gf.Match(np.zeros(3), np.zeros((0, 2), np.float32))
    assert e.value.code == -6                                                      # empty cloud: code, not a CHECK failure
    # points far outside the grid: every lookup is kMinProbability
    r = gf.Match(np.zeros(3), np.array([[30.0, 30.0], [31.0, 29.0]], np.float32))
    assert r.best[1:] == (0, 0) and abs(r.score - 0.1) < 1e-6
    with pytest.raises(RgridError) as e:                                           # more rotated scans than the handle holds
        gf.Match(np.zeros(3), np.array([[5000.0, 5000.0]], np.float32))
    assert e.value.code == -4


def test_insert_matches_oracle_cell_for_cell_and_feeds_the_matcher(gf, oracle_lib):
    """ProbabilityGridRangeDataInserter2D::Insert on the resident grid: three scans from different poses into an
    initially unknown grid, every cell equal to the oracle's after each insertion; then the matcher on that grid."""
    from oracle.binding import oracle_insert, oracle_match
    from reflector_ekf_slam_amd.grid import RangeDataInserterOptions
    _, max_xy, occ = room_grid()
    res = 0.05
    cells = np.zeros((480, 480), np.uint16)
    gf.SetGrid(cells, res, max_xy)
    ref = cells.copy()
    rng = np.random.default_rng(21)
    for k, pose in enumerate(((0.5, 0.3, 0.2), (1.5, -0.8, 1.1), (-2.0, 1.0, -2.0))):
        loc = scan_of(occ, pose, n_points=1500, seed=40 + k)
        c, s = math.cos(pose[2]), math.sin(pose[2])
        world = np.stack([pose[0] + c * loc[:, 0] - s * loc[:, 1], pose[1] + s * loc[:, 0] + c * loc[:, 1]], 1).astype(np.float32)
        ang = rng.uniform(-math.pi, math.pi, 60)
        misses = np.stack([pose[0] + 5.0 * np.cos(ang), pose[1] + 3.5 * np.sin(ang)], 1).astype(np.float32)   # rays that end in free space
        origin = np.array(pose[:2], np.float32)
        gf.Insert(origin, world, misses)
        ref = oracle_insert(ref, res, max_xy, origin, world, misses)
        got = gf.GetGrid()
        assert np.array_equal(got, ref), f"insertion {k}: {np.count_nonzero(got != ref)} cells differ"
        assert (got < 32768).all() and np.count_nonzero(got) > 10000
    # hits only, other probabilities
    gf.Insert(np.zeros(2, np.float32), world[:200], None, RangeDataInserterOptions(False, 0.7, 0.4))
    ref = oracle_insert(ref, res, max_xy, np.zeros(2, np.float32), world[:200], None, 0.7, 0.4, False)
    assert np.array_equal(gf.GetGrid(), ref)
    # a point outside the grid: error code, grid untouched
    from reflector_ekf_slam_amd.grid import RgridError
    with pytest.raises(RgridError) as e:
        gf.Insert(np.zeros(2, np.float32), np.array([[100.0, 0.0]], np.float32), grow=False)
    assert e.value.code == -4 and np.array_equal(gf.GetGrid(), ref)
    with pytest.raises(RgridError) as e:                                      # growing that far exceeds max_cells
        gf.Insert(np.zeros(2, np.float32), np.array([[100.0, 0.0]], np.float32))
    assert e.value.code == -4 and np.array_equal(gf.GetGrid(), ref)
    # the matcher runs on the grid the inserter built
    true = np.array([0.2, 0.1, 0.4])
    pts = scan_of(occ, true, n_points=600, seed=77)
    r = gf.Match(true + [0.1, 0.05, 0.05], pts)
    score, pose, best, info = oracle_match(true + [0.1, 0.05, 0.05], pts, ref, res, max_xy)
    assert r.best == best and abs(r.score - score) <= 1.2e-7 * score
    assert np.abs(r.pose_estimate[:2] - true[:2]).max() <= 0.1


def test_insert_exact_corner_crossings(gf, oracle_lib):
    """Rays between cell CENTRES pass exactly through pixel corners (sub_y == denominator in RayToPixelMask): the
    closed-form column ranges of kg_rays must make the reference's choice there.  Every (dx, dy) cell offset in a
    25 x 25 neighbourhood, from three origins, against the oracle's sequential walk."""
    from oracle.binding import oracle_insert
    res, n = 0.1, 120
    max_xy = (6.0, 6.0)
    cells = np.zeros((n, n), np.uint16)
    k = np.arange(-12, 13)
    for origin_cell in ((60, 60), (30, 75), (90, 20)):
        # centre of cell (row r, col c): x = max_x - (r + 0.5) res, y = max_y - (c + 0.5) res  (map_limits.h:57-62)
        ox, oy = max_xy[0] - (origin_cell[0] + 0.5) * res, max_xy[1] - (origin_cell[1] + 0.5) * res
        gx, gy = np.meshgrid(k, k, indexing="ij")
        ret = np.stack([ox + gx.ravel() * res, oy + gy.ravel() * res], 1).astype(np.float32)
        origin = np.array([ox, oy], np.float32)
        for free in (True,):
            gf.SetGrid(cells, res, max_xy)
            gf.Insert(origin, ret[:300], ret[300:])
            ref = oracle_insert(cells, res, max_xy, origin, ret[:300], ret[300:])
            got = gf.GetGrid()
            assert np.array_equal(got, ref), f"origin {origin_cell}: {np.count_nonzero(got != ref)} cells differ"
    # half-cell offsets: rays that start / end exactly on pixel borders
    ox, oy = max_xy[0] - 60 * res, max_xy[1] - 60 * res
    ret = np.stack([ox + (gx.ravel() + 0.5) * res, oy + gy.ravel() * res], 1).astype(np.float32)
    gf.SetGrid(cells, res, max_xy)
    gf.Insert(np.array([ox, oy], np.float32), ret)
    assert np.array_equal(gf.GetGrid(), oracle_insert(cells, res, max_xy, np.array([ox, oy], np.float32), ret))


def test_grow_as_needed_matches_oracle_and_insert_continues_on_the_grown_grid(gf, oracle_lib):
    """GrowAsNeeded / Grid2D::GrowLimits: a small map, scans that leave it on each side in turn (one, then two
    doublings), odd cell counts; limits, offsets and every cell equal to the oracle's, then Insert on the grown grid."""
    from oracle.binding import oracle_grow, oracle_insert
    res = 0.05
    for (ny, nx) in ((40, 40), (33, 57)):
        max_xy = (1.0, 1.4)
        rng = np.random.default_rng(nx)
        ref = rng.integers(1, 32767, (ny, nx)).astype(np.uint16)
        ref[rng.random((ny, nx)) < 0.3] = 0
        gf.SetGrid(ref, res, max_xy)
        origin = np.array([0.3, 0.5], np.float32)
        for k, far in enumerate(((0.9, 0.8), (2.2, 0.4), (-1.9, 0.2), (0.1, -6.5), (0.2, 7.9))):
            ang = rng.uniform(-math.pi, math.pi, 200)
            rad = rng.uniform(0.05, 0.4, 200)
            ret = np.stack([origin[0] + rad * np.cos(ang), origin[1] + rad * np.sin(ang)], 1).astype(np.float32)
            ret[0] = far
            mis = np.array([[far[0] * 0.5, far[1] * 0.5]], np.float32)
            grown, new_max, off = oracle_grow(ref, res, max_xy, origin, ret, mis)
            gf.GrowAsNeeded(origin, ret, mis)
            lim = gf.GetLimits()
            assert (lim[0], lim[1]) == (grown.shape[1], grown.shape[0]) and lim[3] == new_max[0] and lim[4] == new_max[1], (k, lim, grown.shape, new_max)
            assert np.array_equal(gf.GetGrid(), grown)
            if k == 0:
                assert grown.shape == ref.shape                                # everything inside: no growth
            gf.Insert(origin, ret, mis)                                        # grows again: a no-op now
            ref = oracle_insert(grown, res, new_max, origin, ret, mis)
            max_xy = new_max
            assert np.array_equal(gf.GetGrid(), ref), f"case {k}: {np.count_nonzero(gf.GetGrid() != ref)} cells differ"
        assert ref.shape[0] >= 8 * ny
    from reflector_ekf_slam_amd.grid import RgridError
    with pytest.raises(RgridError) as e:
        gf.GrowAsNeeded(origin, np.array([[np.nan, 0.0]], np.float32))
    assert e.value.code == -1


def test_refine_match_follows_the_oracle_iterate_for_iterate(gf, oracle_lib):
    """CeresScanMatcher2D::Match restated: the one-workgroup LM solve against the oracle's -- same number of iterations,
    same termination, pose equal to 1e-8 (measured: <= 3e-11; the reductions run in another order; the trust-region decisions do not flip).
    The map is built by the inserter, the start pose comes from the correlative matcher, as in MapBuilder::ScanMatch."""
    from oracle.binding import oracle_insert, oracle_refine_match
    from reflector_ekf_slam_amd.grid import CeresScanMatcherOptions2D
    _, max_xy, occ = room_grid()
    res = 0.05
    cells = np.zeros((480, 480), np.uint16)
    for k, pose in enumerate(((0.0, 0.0, 0.0), (1.0, -0.5, 0.7), (-1.5, 0.8, -1.2))):
        loc = scan_of(occ, pose, n_points=1200, seed=60 + k)
        c, s = math.cos(pose[2]), math.sin(pose[2])
        world = np.stack([pose[0] + c * loc[:, 0] - s * loc[:, 1], pose[1] + s * loc[:, 0] + c * loc[:, 1]], 1).astype(np.float32)
        cells = oracle_insert(cells, res, max_xy, np.array(pose[:2], np.float32), world)
    gf.SetGrid(cells, res, max_xy)
    rng = np.random.default_rng(5)
    for trial in range(6):
        true = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.0, 1.0), rng.uniform(-3.0, 3.0)])
        pts = scan_of(occ, true, n_points=(3, 64, 500, 700, 2400, 5000)[trial], seed=100 + trial).astype(np.float32)
        prediction = true + rng.uniform(-1, 1, 3) * [0.08, 0.08, 0.04]
        coarse = gf.Match(prediction, pts).pose_estimate
        for opt in (CeresScanMatcherOptions2D(), CeresScanMatcherOptions2D(2.0, 10.0, 40.0, 100, False), CeresScanMatcherOptions2D(max_num_iterations=3)):
            r = gf.RefineMatch(prediction[:2], coarse, pts, opt)
            pose, summ = oracle_refine_match(prediction[:2], coarse, pts, cells, res, max_xy, opt.occupied_space_weight, opt.translation_weight,
                                             opt.rotation_weight, opt.max_num_iterations, opt.use_nonmonotonic_steps)
            assert (r.iterations, r.termination) == (summ["iterations"], summ["termination"]), (trial, opt, r, summ)
            tol = 1e-8 if len(pts) >= 64 else 1e-5       # 3 points: a nearly flat valley, 70+ iterations, rounding is amplified
            assert np.abs(r.pose_estimate - pose).max() < tol, (trial, r.pose_estimate - pose)
            assert r.initial_cost == pytest.approx(summ["initial_cost"], rel=1e-12) and r.final_cost == pytest.approx(summ["final_cost"], rel=tol)
            assert r.final_cost <= r.initial_cost
        if len(pts) >= 500:
            r = gf.RefineMatch(prediction[:2], coarse, pts)
            assert np.abs(r.pose_estimate[:2] - true[:2]).max() < 0.03 and abs(r.pose_estimate[2] - true[2]) < 0.01
    # a cloud entirely outside the grid: every cell reads kMaxCorrespondenceCost, the gradient of the occupied-space term is 0
    far = np.array([[500.0, 500.0], [501.0, 500.0]], np.float32)
    r = gf.RefineMatch([0.0, 0.0], [0.0, 0.0, 0.0], far)
    pose, summ = oracle_refine_match([0.0, 0.0], [0.0, 0.0, 0.0], far, cells, res, max_xy)
    assert (r.iterations, r.termination) == (summ["iterations"], summ["termination"]) and np.abs(r.pose_estimate - pose).max() < 1e-12
    from reflector_ekf_slam_amd.grid import RgridError
    with pytest.raises(RgridError) as e:
        gf.RefineMatch([0, 0], [0, 0, 0], np.zeros((0, 2), np.float32))
    assert e.value.code == -6


def test_draw_texture_matches_oracle(gf, oracle_lib):
    """ProbabilityGrid::DrawToSubmapTexture: known-cells window, (value, alpha) bytes, slice corner -- byte for byte."""
    from oracle.binding import oracle_draw_texture, oracle_insert
    res = 0.05
    max_xy = (3.0, 4.0)
    empty = np.zeros((130, 170), np.uint16)
    gf.SetGrid(empty, res, max_xy)
    tex, box, sm = gf.DrawTexture()
    otex, obox, osm = oracle_draw_texture(empty, res, max_xy)
    assert box == obox == (0, 0, 1, 1) and sm == osm and np.array_equal(tex, otex) and tex.tolist() == [[[0, 0]]]
    rng = np.random.default_rng(8)
    cells = empty.copy()
    origin = np.array([0.5, 1.0], np.float32)
    for k in range(3):
        ang = rng.uniform(-math.pi, math.pi, 300)
        rad = rng.uniform(0.3, 1.6, 300)
        ret = np.stack([origin[0] + rad * np.cos(ang), origin[1] + rad * np.sin(ang)], 1).astype(np.float32)
        cells = oracle_insert(cells, res, max_xy, origin, ret)
    every = np.arange(170 * 130, dtype=np.uint32).reshape(130, 170)                # every cell value once: the whole table
    for g in (cells, (every % 32768).astype(np.uint16), np.where(every % 7 == 0, 1 + every % 32767, 0).astype(np.uint16)):
        gf.SetGrid(g, res, max_xy)
        tex, box, sm = gf.DrawTexture()
        otex, obox, osm = oracle_draw_texture(g, res, max_xy)
        assert box == obox and sm == osm and np.array_equal(tex, otex)
    big = np.zeros((256, 256), np.uint16)
    big[:, :128] = np.arange(256 * 128).reshape(256, 128) % 32768
    gf.SetGrid(big, res, max_xy)
    tex, box, sm = gf.DrawTexture()
    otex, obox, osm = oracle_draw_texture(big, res, max_xy)
    assert box == obox and np.array_equal(tex, otex)
    # premultiplied alpha: a cell has a value or an alpha, never both
    assert not np.any((tex[..., 0] > 0) & (tex[..., 1] > 0))
