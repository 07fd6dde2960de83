"""tests/witness/grid_witness.py -- TEST INFRASTRUCTURE: second, independent statements of two grid-mapper pieces whose
only other restatement is oracle/grid_oracle.c (reference: src/mapping/probability_grid_range_data_inserter_2d.cc:40-114
with ray_to_pixel_mask.cc:17-168, probability_values.cc; src/scan_matching/real_time_correlative_scan_matcher_2d.cc:20-136).

  * insert_witness: the ray mask is stated GEOMETRICALLY with exact rational arithmetic -- "column X of the pixel grid is
    crossed between the ordinates y_in and y_out; with half-open pixels [Y, Y+1) an ascending ray covers rows floor(y_in) ..
    ceil(y_out) - 1" -- instead of the oracle's (and the reference's) incremental sub-pixel recurrence; the lookup tables are
    rebuilt with vectorised float32 numpy.
  * match_witness: every (rotation, x, y) candidate scored at once with array indexing, the float32 point-order sum as a
    cumulative sum; the oracle loops candidate by candidate.
Neither pins parity with the reference (no tests there, Ceres/Eigen semantics restated): they pin the oracle against a
differently structured implementation.
"""
from __future__ import annotations

import ctypes
import math
from fractions import Fraction

import numpy as np

f32 = np.float32
S = 1000                       # kSubpixelScale (probability_grid_range_data_inserter_2d.cc:16)
MARK = 32768                   # kUpdateMarker (probability_values.h:34)
_libm = ctypes.CDLL("libm.so.6")
for _n in ("cosf", "sinf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]
_libm.atan2f.restype = ctypes.c_float
_libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]


# ---- probability <-> value (probability_values.{h,cc}) --------------------------------------------------------------
def value_to_cost(v):
    v = np.asarray(v, np.int64) & 32767
    lower, upper = f32(1) - (f32(1) - f32(0.1)), f32(1) - f32(0.1)
    k = f32((upper - lower) / (f32(32768) - f32(2)))
    out = (v.astype(f32) * k + f32(lower - k)).astype(f32)
    return np.where(v == 0, upper, out).astype(f32)


def cost_to_value(c):
    lower, upper = f32(1) - (f32(1) - f32(0.1)), f32(1) - f32(0.1)
    cl = np.minimum(np.maximum(np.asarray(c, f32), lower), upper).astype(f32)
    x = ((cl - lower).astype(f32) * f32(f32(32766) / f32(upper - lower))).astype(f32)
    return (np.floor(x.astype(np.float64) + 0.5).astype(np.int64) + 1)          # lroundf of a non-negative float


def lookup_table(probability):
    """ComputeLookupTableToApplyCorrespondenceCostOdds (probability_values.cc:76-96), marker set."""
    p = f32(probability)
    odds = f32(p / f32(f32(1) - p))
    t = np.zeros(32768, np.int64)
    t[0] = cost_to_value(f32(f32(1) - f32(odds / f32(odds + f32(1))))) + MARK
    cells = np.arange(1, 32768)
    pc = (f32(1) - value_to_cost(cells)).astype(f32)
    o = (odds * (pc / (f32(1) - pc).astype(f32)).astype(f32)).astype(f32)
    pr = (o / (o + f32(1)).astype(f32)).astype(f32)
    t[1:] = cost_to_value((f32(1) - pr).astype(f32)) + MARK
    return t


# ---- ray mask, geometrically ----------------------------------------------------------------------------------------
def ray_pixels(bx, by, ex, ey):
    """Pixels (X, Y) touched by the segment between the CENTRES of sub-pixels (bx, by) and (ex, ey) (super-scaled indices,
    kSubpixelScale sub-pixels per pixel), half-open pixels.  Coordinates are doubled so that centres are integers:
    centre = 2 i + 1, pixel boundaries at multiples of 2 S."""
    if bx > ex:
        bx, by, ex, ey = ex, ey, bx, by
    X0, X1 = bx // S, ex // S
    if X0 == X1:
        y0, y1 = min(by, ey) // S, max(by, ey) // S
        return [(X0, y) for y in range(y0, y1 + 1)]
    x0, y0, x1, y1 = 2 * bx + 1, 2 * by + 1, 2 * ex + 1, 2 * ey + 1
    slope = Fraction(y1 - y0, x1 - x0)
    D = 2 * S
    out = []
    for X in range(X0, X1 + 1):
        xa = max(x0, X * D)
        xb = min(x1, (X + 1) * D)
        ya = Fraction(y0) + slope * (xa - x0)
        yb = Fraction(y0) + slope * (xb - x0)
        if slope > 0:
            lo, hi = math.floor(ya / D), math.ceil(yb / D) - 1
        elif slope < 0:
            lo, hi = math.floor(yb / D), math.ceil(ya / D) - 1
        else:
            lo = hi = math.floor(ya / D)
        # a segment END sits on a sub-pixel centre, never on a pixel boundary: its pixel is always inside [lo, hi]
        out.extend((X, Y) for Y in range(lo, hi + 1))
    return out


def _cell_index(px, py, max_x, max_y, res):
    """MapLimits::GetCellIndex (map_limits.h:48-57): x index from y."""
    rnd = lambda v: int(math.floor(v + 0.5)) if v >= 0 else -int(math.floor(-v + 0.5))       # lround
    return rnd((max_y - float(py)) / res - 0.5), rnd((max_x - float(px)) / res - 0.5)


def insert_witness(cells, resolution, max_xy, origin, returns_xy, misses_xy=None, hit_probability=0.55,
                   miss_probability=0.49, insert_free_space=True):
    """ProbabilityGridRangeDataInserter2D::Insert on cells[ny, nx] (uint16).  Returns the new grid, or None when a point
    falls outside (the oracle reports -1 and leaves the grid alone; growing is a separate step)."""
    g = np.array(cells, np.int64)
    ny, nx = g.shape
    rs = resolution / S
    pts = [tuple(origin)] + [tuple(p) for p in np.asarray(returns_xy, f32).reshape(-1, 2)] + \
          [tuple(p) for p in (np.asarray(misses_xy, f32).reshape(-1, 2) if misses_xy is not None else [])]
    idx = [_cell_index(p[0], p[1], max_xy[0], max_xy[1], rs) for p in pts]
    if any(ix < 0 or iy < 0 or ix >= nx * S or iy >= ny * S for ix, iy in idx):
        return None
    n_ret = np.asarray(returns_xy).reshape(-1, 2).shape[0]
    hit, miss = lookup_table(hit_probability), lookup_table(miss_probability)
    for ix, iy in idx[1:1 + n_ret]:                             # hits first (:57-62)
        x, y = ix // S, iy // S
        if g[y, x] < MARK:
            g[y, x] = hit[g[y, x]]
    if insert_free_space:
        bx, by = idx[0]
        for ix, iy in idx[1:]:                                  # then the rays to every return and miss (:69-91)
            for x, y in ray_pixels(bx, by, ix, iy):
                if g[y, x] < MARK:
                    g[y, x] = miss[g[y, x]]
    g[g >= MARK] -= MARK                                        # FinishUpdate (grid_2d.cc:20-29)
    return g.astype(np.uint16)


# ---- real-time correlative matcher ----------------------------------------------------------------------------------
def _rotation_cs(angle):
    """Project2D(Rigid3f::Rotation(AngleAxisf(angle, UnitZ))) as a float32 (cos, sin) (transform.h:27-41,93-98)."""
    ha = f32(f32(0.5) * f32(angle))
    w, z = f32(_libm.cosf(float(ha))), f32(_libm.sinf(float(ha)))
    uvy = f32(z + z)
    dx = f32(f32(f32(1) + f32(w * f32(0))) + f32(f32(f32(0) * f32(0)) - f32(z * uvy)))
    dy = f32(f32(f32(0) + f32(w * uvy)) + f32(f32(z * f32(0)) - f32(f32(0) * f32(0))))
    yaw = f32(_libm.atan2f(float(dy), float(dx)))
    return f32(_libm.cosf(float(yaw))), f32(_libm.sinf(float(yaw)))


def _rotate(pts, c, s):
    x, y = pts[:, 0], pts[:, 1]
    return np.stack([((c * x).astype(f32) - (s * y).astype(f32)).astype(f32), ((s * x).astype(f32) + (c * y).astype(f32)).astype(f32)], -1)


def match_witness(initial_pose, points_xy, cells, resolution, max_xy, linear_search_window=0.2, angular_search_window=0.26,
                  translation_delta_cost_weight=1e-1, rotation_delta_cost_weight=1e-1):
    """RealTimeCorrelativeScanMatcher2D::Match.  -> (score float32, pose (3,), (scan, x_off, y_off))."""
    pts = np.ascontiguousarray(points_xy, f32).reshape(-1, 2)
    n = pts.shape[0]
    g = np.asarray(cells, np.int64)
    ny, nx = g.shape
    rot0 = _rotate(pts, *_rotation_cs(f32(initial_pose[2])))
    rng = np.sqrt(((rot0[:, 0] * rot0[:, 0]).astype(f32) + (rot0[:, 1] * rot0[:, 1]).astype(f32)).astype(f32)).astype(f32)
    max_range = max(f32(f32(3) * f32(resolution)), rng.max() if n else f32(0))
    step = (1. - 1e-3) * math.acos(1. - (resolution * resolution) / (2. * float(f32(max_range * max_range))))
    na = int(math.ceil(angular_search_window / step))
    nl = int(math.ceil(linear_search_window / resolution))
    prob = np.concatenate([(f32(1) - value_to_cost(np.arange(32768))).astype(f32)] * 2)       # value (with or without marker) -> probability
    tx, ty = f32(initial_pose[0]), f32(initial_pose[1])
    offs = np.arange(-nl, nl + 1)
    best = None
    dth = -na * step
    for scan in range(2 * na + 1):
        rot = _rotate(rot0, *_rotation_cs(f32(dth)))
        dth += step
        px, py = (rot[:, 0] + tx).astype(f32), (rot[:, 1] + ty).astype(f32)
        lround = lambda v: np.where(v >= 0, np.floor(v + 0.5), -np.floor(-v + 0.5)).astype(np.int64)
        ix = lround((max_xy[1] - py.astype(np.float64)) / resolution - 0.5)
        iy = lround((max_xy[0] - px.astype(np.float64)) / resolution - 0.5)
        cx = ix[None, None, :] + offs[:, None, None]            # [x_off, y_off, point]
        cy = iy[None, None, :] + offs[None, :, None]
        inside = (cx >= 0) & (cy >= 0) & (cx < nx) & (cy < ny)
        p = np.where(inside, prob[g[np.clip(cy, 0, ny - 1), np.clip(cx, 0, nx - 1)]], f32(0.1)).astype(f32)
        score = (np.add.accumulate(p, axis=2, dtype=f32)[:, :, -1] / f32(n)).astype(f32)     # float32 sum in point order
        orientation = (scan - na) * step
        x = -offs[None, :] * resolution + 0.0 * offs[:, None]
        y = -offs[:, None] * resolution + 0.0 * offs[None, :]
        a = np.hypot(x, y) * translation_delta_cost_weight + abs(orientation) * rotation_delta_cost_weight
        score = (score.astype(np.float64) * np.exp(-(a * a))).astype(f32)
        k = int(np.argmax(score))                               # first maximum in (x_off, y_off) order
        xi, yi = divmod(k, offs.size)
        if best is None or score[xi, yi] > best[0]:
            best = (score[xi, yi], (initial_pose[0] + x[xi, yi], initial_pose[1] + y[xi, yi], initial_pose[2] + orientation),
                    (scan, int(offs[xi]), int(offs[yi])))
    return best
