"""CPU tests of the grid front-end ORACLE (oracle/grid_oracle.c) against independent numpy restatements and
hand-checkable cases.  The reference ships no tests for this code either: parity unpinned (see the oracle header)."""
from __future__ import annotations

import math

import numpy as np
import pytest

from tests.grid_cases import room_grid, scan_of


def _np_voxel_filter(pts, res):
    seen, out = set(), []
    for p in np.asarray(pts, np.float32):
        # RoundToInt = lround: half away from zero (port.h:25); float32 division
        q = np.float32(p) / np.float32(res)
        k = tuple(int(math.floor(abs(float(v)) + 0.5) * (1 if v >= 0 else -1)) for v in q)
        if k not in seen:
            seen.add(k); out.append(p)
    return np.array(out, np.float32).reshape(-1, 2)


def test_voxel_filter_keeps_first_point_per_voxel_in_order(oracle_lib):
    from oracle.binding import oracle_voxel_filter
    rng = np.random.default_rng(0)
    pts = rng.uniform(-6, 6, (3000, 2)).astype(np.float32)
    for res in (0.025, 0.3, 0.9):
        got = oracle_voxel_filter(pts, res)
        assert np.array_equal(got, _np_voxel_filter(pts, res))
    # hand case: half-way values round away from zero; the second point of a voxel is dropped
    hand = np.array([[0.5, 0.5], [0.49, 0.2], [1.4, -0.5], [0.6, 1.4], [-0.5, 0.0], [-1.49, 0.4]], np.float32)
    assert np.array_equal(oracle_voxel_filter(hand, 1.0), hand[[0, 1, 2, 4]])
    assert oracle_voxel_filter(np.zeros((0, 2), np.float32), 0.1).shape == (0, 2)


def test_adaptive_voxel_filter_paths(oracle_lib):
    from oracle.binding import oracle_adaptive_voxel_filter, oracle_voxel_filter
    rng = np.random.default_rng(1)
    pts = rng.uniform(-8, 8, (4000, 2)).astype(np.float32)
    # (a) already sparse: returned unchanged apart from the range gate
    few = pts[:300]
    assert np.array_equal(oracle_adaptive_voxel_filter(few, 0.9, 500, 100.0), few)
    gated = oracle_adaptive_voxel_filter(few, 0.9, 500, 5.0)
    assert np.array_equal(gated, few[np.sqrt(few[:, 0] ** 2 + few[:, 1] ** 2) <= np.float32(5.0)])
    # (b) max_length already leaves enough points
    assert np.array_equal(oracle_adaptive_voxel_filter(pts, 0.3, 500, 100.0), oracle_voxel_filter(pts, 0.3))
    # (c) bisection: the result has >= min points, a voxel size 10 % larger would not (within the bisection tolerance)
    out = oracle_adaptive_voxel_filter(pts, 2.0, 500, 100.0)
    assert 500 <= out.shape[0] < 700
    assert oracle_voxel_filter(pts, 2.0).shape[0] < 500
    # (d) impossible target: gives up after shrinking the edge by 1e-2 and returns the last attempt
    dense = np.tile(np.array([[1.0, 1.0]], np.float32), (600, 1)) + rng.normal(0, 1e-4, (600, 2)).astype(np.float32)
    out = oracle_adaptive_voxel_filter(dense, 0.9, 550, 100.0)
    assert 1 <= out.shape[0] < 550


def test_value_to_probability_table(oracle_lib):
    L = oracle_lib
    f32 = np.float32
    lower, upper = f32(1) - (f32(1) - f32(0.1)), f32(1) - f32(0.1)
    kscale = (upper - lower) / f32(32766)
    for v in (0, 1, 2, 1000, 16384, 32767, 32768, 32769, 40000, 65535):
        vv = v & 32767
        cost = upper if vv == 0 else f32(vv) * kscale + (lower - kscale)
        assert L.ogrid_value_to_probability(v) == float(f32(1) - cost)
    assert abs(L.ogrid_value_to_probability(1) - 0.9) < 1e-6 and abs(L.ogrid_value_to_probability(32767) - 0.1) < 1e-6


def test_matcher_recovers_a_known_offset_and_obeys_the_search_structure(oracle_lib):
    from oracle.binding import oracle_match
    cells, max_xy, occ = room_grid()
    true = np.array([0.8, -0.6, 0.35])
    pts = scan_of(occ, true)
    init = true + np.array([0.10, -0.15, math.radians(4.0)])
    score, pose, best, info = oracle_match(init, pts, cells, 0.05, max_xy)
    num_scans, num_linear, ncand = info
    assert num_linear == 4 and ncand == num_scans * 81 and num_scans % 2 == 1           # ceil(0.2 / 0.05); 2 n + 1 scans
    # angular step: (1 - 1e-3) acos(1 - r^2 / (2 dmax^2)) with dmax the largest range of the rotated cloud
    dmax = float(np.sqrt((pts.astype(np.float64) ** 2).sum(1)).max())
    step = (1 - 1e-3) * math.acos(1 - 0.05 ** 2 / (2 * dmax ** 2))
    assert abs((num_scans - 1) / 2 - math.ceil(math.radians(15.0) / step)) <= 1
    assert np.abs(pose[:2] - true[:2]).max() <= 0.05 + 1e-9 and abs(pose[2] - true[2]) < 2.5 * step
    assert 0.5 < score < 0.9
    # pose_estimate is the initial pose plus the candidate's offsets: x = -y_off * r, y = -x_off * r (Candidate2D)
    scan, xo, yo = best
    assert np.allclose(pose, init + np.array([-yo * 0.05, -xo * 0.05, (scan - (num_scans - 1) // 2) * step]), atol=1e-6)
    # a perfect initial guess is its own best candidate: zero offsets win through the exp(-(.)^2) penalty
    _, _, best0, info0 = oracle_match(true, scan_of(occ, true, noise=0.0), cells, 0.05, max_xy)
    assert best0[1] == 0 and best0[2] == 0 and abs(best0[0] - (info0[0] - 1) // 2) <= 1


def test_matcher_outside_the_grid_scores_min_probability(oracle_lib):
    from oracle.binding import oracle_match
    cells = np.full((40, 40), 2000, np.uint16)
    pts = np.array([[50.0, 50.0], [51.0, 49.0], [52.0, 50.5]], np.float32)                # far outside: kMinProbability everywhere
    score, pose, best, info = oracle_match(np.zeros(3), pts, cells, 0.05, (1.0, 1.0), linear_search_window=0.1)
    assert best[1] == 0 and best[2] == 0                                                   # all sums equal: the smallest penalty wins
    assert abs(score - 0.1) < 1e-6


def test_lookup_tables_and_hit_miss_semantics(oracle_lib):
    from oracle.binding import oracle_insert, oracle_lookup_table
    hit, miss = oracle_lookup_table(0.55), oracle_lookup_table(0.49)
    assert hit.min() >= 32768 and miss.min() >= 32768                            # every entry carries the update marker
    f32 = np.float32
    # unknown cell: value of the probability itself (probability_values.cc:80-82): cost = 1 - p
    lower, upper = f32(1) - (f32(1) - f32(0.1)), f32(1) - f32(0.1)
    v0 = int(np.rint((f32(1) - f32(0.55) - lower) * (f32(32766) / (upper - lower)))) + 1
    assert abs(int(hit[0]) - 32768 - v0) <= 1
    # a hit lowers the correspondence cost of a known cell, a miss raises it; both are monotone in the old value
    known = np.arange(1, 32768)
    assert ((hit[known] - 32768).astype(int) <= known).all() and ((miss[known] - 32768).astype(int) >= known).all()
    assert (np.diff((hit[known] - 32768).astype(int)) >= 0).all()
    # one return seen from the origin: its cell gets the hit table, the cells on the ray the miss table, hits win
    n = 80
    cells = np.zeros((n, n), np.uint16)
    res, max_xy = 0.1, (4.0, 4.0)
    origin = np.array([0.03, -0.02], np.float32)
    ret = np.array([[2.51, 1.27], [2.65, 1.15]], np.float32)                    # second return: same ray region, other cell
    out = oracle_insert(cells, res, max_xy, origin, ret)
    def cell_of(p):
        return int(np.floor((max_xy[0] - p[0]) / res)), int(np.floor((max_xy[1] - p[1]) / res))   # (row = y index from x, col = x index from y)
    for p in ret:
        r, c = cell_of(p)
        assert out[r, c] == hit[0] - 32768
    r0, c0 = cell_of(origin)
    assert out[r0, c0] == miss[0] - 32768
    touched = np.argwhere(out != 0)
    assert 30 < touched.shape[0] < 90                                            # ~ |dx| + |dy| cells per ray, two nearly equal rays
    assert (out[out != 0] < 32768).all()                                         # FinishUpdate removed every marker
    # supercover property: every point of the segment lies in an updated cell
    for p in ret:
        for t in np.linspace(0, 1, 2001):
            q = origin.astype(np.float64) * (1 - t) + p.astype(np.float64) * t
            r, c = cell_of(q)
            assert out[r, c] != 0
    # a second identical insertion moves known cells: hit cell gets more certain, ray cells freer
    out2 = oracle_insert(out, res, max_xy, origin, ret)
    r, c = cell_of(ret[0])
    assert out2[r, c] < out[r, c] and out2[r0, c0] > out[r0, c0]
    # hits only
    out3 = oracle_insert(cells, res, max_xy, origin, ret, insert_free_space=False)
    assert np.count_nonzero(out3) == 2
    with pytest.raises(ValueError):
        oracle_insert(cells, res, max_xy, origin, np.array([[9.0, 0.0]], np.float32))


def test_grow_limits_doubles_around_the_old_grid(oracle_lib):
    """Grid2D::GrowLimits (grid_2d.cc:59-99): the grid doubles until the padded bounding box is inside; the old cells
    sit at the returned offset, everything else is unknown, and cell centres keep their world coordinates."""
    from oracle.binding import oracle_grow
    res = 0.1
    rng = np.random.default_rng(2)
    cells = rng.integers(1, 32767, (21, 30)).astype(np.uint16)                 # (num_y_cells, num_x_cells)
    max_xy = (1.5, 2.0)
    origin = np.array([0.2, 0.3], np.float32)
    inside = np.array([[0.5, 0.5], [-0.3, 0.1]], np.float32)
    g, mx, off = oracle_grow(cells, res, max_xy, origin, inside)
    assert g.shape == cells.shape and mx == max_xy and off == (0, 0) and np.array_equal(g, cells)
    far = np.array([[4.0, 0.0]], np.float32)                                   # beyond max_x: two doublings
    g, mx, off = oracle_grow(cells, res, max_xy, origin, inside, far)
    assert g.shape == (84, 120)
    assert off == (15 + 30, 10 + 21)                                           # nx/2 + (2 nx)/2 , ny/2 + (2 ny)/2
    assert mx[0] == pytest.approx(1.5 + res * (10 + 21)) and mx[1] == pytest.approx(2.0 + res * (15 + 30))
    assert np.array_equal(g[off[1]:off[1] + 21, off[0]:off[0] + 30], cells)
    assert np.count_nonzero(g) == np.count_nonzero(cells)
    # the centre of old cell (row r, col c) is unchanged: x = max_x - (r + 0.5) res, y = max_y - (c + 0.5) res
    assert mx[0] - (off[1] + 0.5) * res == pytest.approx(max_xy[0] - 0.5 * res)
    assert mx[1] - (off[0] + 0.5) * res == pytest.approx(max_xy[1] - 0.5 * res)
    # the far point is now inside
    ix = round((mx[1] - 0.0) / res - 0.5); iy = round((mx[0] - 4.0) / res - 0.5)
    assert 0 <= ix < g.shape[1] and 0 <= iy < g.shape[0]


def _np_refine_residuals(p, pts, cells, res, max_xy, target, angle0, w=(1.0, 0.1, 0.4)):
    """Independent numpy statement of the three residual blocks (vectorised Catmull-Rom bicubic)."""
    ny, nx = cells.shape
    v = (cells & 32767).astype(np.float32)
    one, kmin = np.float32(1), np.float32(0.1)                                  # float32 constants as the reference forms them
    upper, lower = one - kmin, one - (one - kmin)
    kscale = (upper - lower) / np.float32(32766)
    cost = np.where(v == 0, upper, v * kscale + (lower - kscale)).astype(np.float64)
    c, s = math.cos(p[2]), math.sin(p[2])
    wx = c * pts[:, 0] - s * pts[:, 1] + p[0]
    wy = s * pts[:, 0] + c * pts[:, 1] + p[1]
    pad = float(2147483647 // 4)                                               # kPadding: costs ~1e-7 cell of resolution
    r = (max_xy[0] - wx) / res - 0.5 + pad
    q = (max_xy[1] - wy) / res - 0.5 + pad
    fr, fq = r - np.floor(r), q - np.floor(q)
    r0, q0 = (np.floor(r) - pad).astype(int), (np.floor(q) - pad).astype(int)

    def at(row, col):
        ok = (row >= 0) & (col >= 0) & (row < ny) & (col < nx)
        return np.where(ok, cost[np.clip(row, 0, ny - 1), np.clip(col, 0, nx - 1)], float(upper))

    def spline(p0, p1, p2, p3, x):
        a = 0.5 * (-p0 + 3 * p1 - 3 * p2 + p3)
        b = 0.5 * (2 * p0 - 5 * p1 + 4 * p2 - p3)
        cc = 0.5 * (-p0 + p2)
        return p1 + x * (cc + x * (b + x * a))

    rows = [spline(at(r0 - 1 + a, q0 - 1), at(r0 - 1 + a, q0), at(r0 - 1 + a, q0 + 1), at(r0 - 1 + a, q0 + 2), fq) for a in range(4)]
    f = spline(rows[0], rows[1], rows[2], rows[3], fr)
    return np.concatenate([w[0] / math.sqrt(len(pts)) * f, [w[1] * (p[0] - target[0]), w[1] * (p[1] - target[1]), w[2] * (p[2] - angle0)]])


def test_refine_match_restatement_against_scipy_least_squares(oracle_lib):
    """CeresScanMatcher2D::Match restated (Ceres is absent: parity unpinned).  Cross-checks: the oracle's cost at the
    start equals an independent numpy statement of the residuals; its answer is a local minimum of that cost and agrees
    with scipy's trust-region least-squares solver started from the same point; it pulls a perturbed pose back."""
    from scipy.optimize import least_squares
    from oracle.binding import oracle_insert, oracle_refine_match
    _, max_xy, occ = room_grid()
    res = 0.05
    cells = np.zeros((480, 480), np.uint16)
    for k, pose in enumerate(((0.0, 0.0, 0.0), (1.0, -0.5, 0.7), (-1.5, 0.8, -1.2))):
        loc = scan_of(occ, pose, n_points=1200, seed=60 + k)
        c, s = math.cos(pose[2]), math.sin(pose[2])
        world = np.stack([pose[0] + c * loc[:, 0] - s * loc[:, 1], pose[1] + s * loc[:, 0] + c * loc[:, 1]], 1).astype(np.float32)
        cells = oracle_insert(cells, res, max_xy, np.array(pose[:2], np.float32), world)
    true = np.array([0.3, -0.2, 0.25])
    pts = scan_of(occ, true, n_points=500, seed=91).astype(np.float32)
    from oracle.binding import oracle_match
    prediction = true + [-0.06, 0.05, -0.03]                                   # more than a cell away: the correlative matcher first,
    _, coarse, _, _ = oracle_match(prediction, pts, cells, res, max_xy)        # as MapBuilder::ScanMatch does (map_builder.cc:43-55)
    for start, target in ((true + [0.04, -0.03, 0.02], None), (np.array(coarse), prediction[:2]), (true, None)):
        target = start[:2] if target is None else target
        pose, summ = oracle_refine_match(target, start, pts, cells, res, max_xy)
        r0 = _np_refine_residuals(start, pts.astype(np.float64), cells, res, max_xy, target, start[2])
        assert summ["initial_cost"] == pytest.approx(0.5 * float(r0 @ r0), rel=1e-12)
        assert summ["termination"] == 0 and 1 <= summ["iterations"] < 100
        assert summ["final_cost"] <= summ["initial_cost"]
        rf = _np_refine_residuals(pose, pts.astype(np.float64), cells, res, max_xy, target, start[2])
        assert summ["final_cost"] == pytest.approx(0.5 * float(rf @ rf), rel=1e-12)
        sol = least_squares(_np_refine_residuals, start, args=(pts.astype(np.float64), cells, res, max_xy, target, start[2]),
                            method="lm", xtol=1e-10, ftol=1e-10, gtol=1e-10, diff_step=1e-4)   # finite differences must step over
        scipy_cost = 0.5 * float(sol.fun @ sol.fun)                                   # the kPadding quantisation of the coordinates
        assert abs(scipy_cost - summ["final_cost"]) <= 2e-3 * scipy_cost             # function_tolerance 1e-6 stops a little short
        assert np.abs(pose - sol.x).max() < 5e-3
        assert np.abs(pose[:2] - true[:2]).max() < 0.03 and abs(pose[2] - true[2]) < 0.01
    # monotonic steps, a hard iteration cap, heavier priors
    p1, s1 = oracle_refine_match(target, true + [0.05, 0.05, 0.03], pts, cells, res, max_xy, use_nonmonotonic_steps=False)
    assert s1["termination"] == 0 and np.abs(p1[:2] - true[:2]).max() < 0.03
    p2, s2 = oracle_refine_match(target, true + [0.05, 0.05, 0.03], pts, cells, res, max_xy, max_num_iterations=1)
    assert s2["termination"] == 1 and s2["iterations"] == 1
    p3, s3 = oracle_refine_match(true[:2] + [0.05, 0.05], true + [0.05, 0.05, 0.03], pts, cells, res, max_xy,
                                 translation_weight=1e3, rotation_weight=1e3)
    assert np.abs(p3 - (true + [0.05, 0.05, 0.03])).max() < 1e-3                       # the priors pin the pose


def test_draw_texture_semantics(oracle_lib):
    """DrawToSubmapTexture (probability_grid.cc:86-131): unknown -> (0, 0); probability 0.5 -> log-odds integer 128 ->
    delta 0 -> (0, 1); occupied cells carry alpha, free cells carry value; the window is the known cells' box."""
    from oracle.binding import oracle_draw_texture
    g = np.zeros((20, 30), np.uint16)
    # cost value v <-> probability 1 - (0.1 + (v - 1) * 0.8 / 32766): v = 1 -> p = 0.9, v = 32767 -> p = 0.1, v = 16384 -> p ~ 0.5
    g[5, 7] = 1; g[9, 12] = 32767; g[6, 8] = 16384
    tex, box, sm = oracle_draw_texture(g, 0.1, (2.0, 3.0))
    assert box == (7, 5, 6, 5) and tex.shape == (5, 6, 2)
    assert sm == (2.0 - 0.1 * 5, 3.0 - 0.1 * 7)
    assert tex[0, 0].tolist() == [0, 127]                                       # p = 0.9: log-odds integer 255, delta -127 -> alpha
    assert tex[4, 5].tolist() == [127, 0]                                       # p = 0.1: integer 1, delta 127 -> value
    assert tex[1, 1].tolist() == [0, 1]                                         # p ~ 0.5: delta 0 -> (0, 1)
    assert tex[2, 2].tolist() == [0, 0]                                         # unknown inside the window
