"""tests/witness/detect2d_witness.py -- TEST INFRASTRUCTURE: a SECOND, independent statement of the reference's 2D
reflector detector (/root/reference/src/reflector_detect/laser/laser_reflector_detect.cc:23-316 with
pose_extrapolator.cc), structured differently from oracle/detect2d_oracle.c:

  * the oracle walks the beams once with the reference's own sequential state machine;
  * here run membership is derived DATA-PARALLEL from numpy arrays -- "previous bright beam", continue / bridge / start
    flags, run ids by cumulative sums -- and only the handful of runs is then gated in a short loop; point geometry is
    evaluated per member afterwards.  (This is also how the GPU kernel decomposes the problem; the two CPU statements
    share no code.)

Scalar float32 transcendentals go through the same libm (cosf / sinf / hypotf via ctypes) so that agreement with the
oracle is exact, not approximate.  Like the oracle this pins nothing against the reference itself (it ships no tests).
"""
from __future__ import annotations

import ctypes
import math

import numpy as np

_libm = ctypes.CDLL("libm.so.6")
for _n in ("cosf", "sinf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]
_libm.hypotf.restype = ctypes.c_float
_libm.hypotf.argtypes = [ctypes.c_float, ctypes.c_float]
f32 = np.float32


def _cosf(a): return f32(_libm.cosf(float(a)))
def _sinf(a): return f32(_libm.sinf(float(a)))
def _hypotf(a, b): return f32(_libm.hypotf(float(a), float(b)))


def _apply_f(r, px, py):
    """transform::Rigid2f * point (rigid_transform.h:87-92), float32 throughout."""
    c, s = _cosf(r[2]), _sinf(r[2])
    return (f32(f32(f32(c * px) + f32(f32(-s) * py)) + r[0]), f32(f32(f32(s * px) + f32(c * py)) + r[1]))


def _cast(r): return (f32(r[0]), f32(r[1]), f32(r[2]))


def _mul_d(l, r):
    c, s = math.cos(l[2]), math.sin(l[2])
    return ((c * r[0] + (-s) * r[1]) + l[0], (s * r[0] + c * r[1]) + l[1], l[2] + r[2])


def _inv_d(r):
    c, s = math.cos(-r[2]), math.sin(-r[2])
    return (-(c * r[0] + (-s) * r[1]), -(s * r[0] + c * r[1]), -r[2])


class Extrapolator:
    """PoseExtrapolator (pose_extrapolator.cc:12-129), default build."""

    def __init__(self):
        self.odom = []          # (time, px, py, qz, qw, vx, vy, wz)

    def add(self, o): self.odom.append(tuple(float(v) for v in o))

    def trim(self, t):
        while len(self.odom) > 1 and self.odom[0][0] < t:
            self.odom.pop(0)

    @staticmethod
    def _interp(st, time):
        t0, px, py, qz, qw, vx, vy, wz = st
        yaw0 = 2 * math.atan2(qz, qw)
        if t0 <= time:
            dt = t0 - time
            yaw = yaw0 - wz * dt
            return (px - vx * dt * math.cos(yaw) + vy * dt * math.sin(yaw),
                    py - vx * dt * math.sin(yaw) - vy * dt * math.cos(yaw), yaw)
        dt = time - t0
        yaw = yaw0 - wz * dt                                   # the reference's sign in the backward branch (:118-128)
        return (px + vx * dt * math.cos(yaw) - vy * dt * math.sin(yaw),
                py + vx * dt * math.sin(yaw) + vy * dt * math.cos(yaw), yaw)

    def pose(self, time):
        if not self.odom:
            return (0.0, 0.0, 0.0)
        if time <= self.odom[0][0]:
            return self._interp(self.odom[0], time)
        return self._interp(self.odom[-1], time)                # t >= back, and the no-break loop in between (:76-82)


def detect2d_witness(scan, extrap: Extrapolator, intensity_min=160.0, reflector_min_length=0.18, reflector_length_error=0.06,
                     det_range_min=0.3, det_range_max=10.0, s2b=(0.0, 0.0, 0.0)):
    """scan: object/dict with the LaserScan fields.  -> (centers (K,2) f32, members: list of beam-index arrays, returns (n,2) f32)."""
    g = (lambda k: scan[k]) if isinstance(scan, dict) else (lambda k: getattr(scan, k))
    ranges = np.ascontiguousarray(g("ranges"), f32)
    inten = np.ascontiguousarray(g("intensities"), f32)
    N = ranges.shape[0]
    amin, amax, inc = f32(g("angle_min")), f32(g("angle_max")), f32(g("angle_increment"))
    scan_time, rmin, rmax, stamp = f32(g("scan_time")), f32(g("range_min")), f32(g("range_max")), float(g("stamp"))
    det_lo, det_hi = f32(det_range_min), f32(det_range_max)
    if rmin < 0 or rmax <= rmin or (inc < 0 and amax <= amin):
        raise ValueError("invalid scan")
    if N == 0:
        return np.zeros((0, 2), f32), [], np.zeros((0, 2), f32)
    dt_pt = float(f32(scan_time / f32(N)))
    t_first = stamp - float(scan_time)
    extrap.trim(t_first)
    s2bf = _cast(tuple(float(v) for v in s2b))
    circle = (float(f32(amax - amin)) - 2 * math.pi) < 1e-6
    # beam angle: float32 accumulation (:51,:175)
    ang = np.add.accumulate(np.concatenate([[amin], np.full(N - 1, inc, f32)]).astype(f32), dtype=f32)
    idx = np.arange(N)
    valid = (ranges >= rmin) & (ranges <= rmax)
    back = np.maximum.accumulate(np.where(valid, idx, -1))      # beam whose point is point_cloud.back() when beam i is handled
    bright = (det_lo <= ranges) & (ranges <= det_hi) & (inten.astype(np.float64) > intensity_min) & (back >= 0)
    bi = np.nonzero(bright)[0]
    t_of = lambda j: f32(t_first + j * dt_pt)                   # stored in a Vector3f (sensor_data.h:18)

    def point_of_valid(j):                                      # the point pushed for valid beam j (:66-72)
        return _apply_f(s2bf, f32(ranges[j] * _cosf(ang[j])), f32(ranges[j] * _sinf(ang[j])))

    members = []                                                # per run: list of (beam id, kind) kind 0 = bright beam, 1 = gap beam
    if bi.size:
        prev = np.concatenate([[-1], bi[:-1]])
        gap = bi - prev
        nxt = np.minimum(bi + 1, N - 1)
        cont = (prev >= 0) & (gap == 1)
        bridge = (prev >= 0) & (gap > 1) & (gap < 4) & \
                 (np.abs((ranges[bi] - ranges[np.maximum(prev, 0)]).astype(f32)).astype(np.float64) < 0.3) & \
                 (inten[nxt].astype(np.float64) > intensity_min)
        start = ~(cont | bridge)
        rid = np.cumsum(start) - 1
        for r in range(rid[-1] + 1):
            sel = np.nonzero(rid == r)[0]
            mem = []
            for k in sel:
                if bridge[k]:
                    for j in range(prev[k] + 1, bi[k]):
                        if not np.isinf(ranges[j]):
                            mem.append((j, 1, bi[k]))
                mem.append((bi[k], 0, bi[k]))
            members.append(mem)

    def run_points(mem):
        pts = []
        for j, kind, host in mem:
            if kind == 0:
                b = back[j]
                x, y = point_of_valid(b)
                pts.append((x, y, t_of(b), j))
            else:                                               # gap beam: its own range at angle - inc * (i - j) (:115-130)
                a = f32(ang[host] - f32(inc * f32(host - j)))
                x, y = _apply_f(s2bf, f32(ranges[j] * _cosf(a)), f32(ranges[j] * _sinf(a)))
                pts.append((x, y, t_of(j), j))
        return pts

    def length_ok(pts):
        ln = _hypotf(f32(pts[0][0] - pts[-1][0]), f32(pts[0][1] - pts[-1][1]))
        return abs(float(ln) - reflector_min_length) < reflector_length_error

    runs = [run_points(m) for m in members]
    clusters, first_ids = [], []
    if runs:
        for r in runs[:-1]:                                     # runs closed by the start of the next one (:140-169)
            if (circle and r[0][3] == 0) or length_ok(r):
                clusters.append(list(r)); first_ids.append(r[0][3])
        cur = runs[-1]                                          # the run still open at the end (:178-236)
        if clusters:
            first_id = first_ids[0] if first_ids else -1
            first_pt, first_last = clusters[0][0], clusters[0][-1]
            d = (f32(cur[-1][0] - first_pt[0]), f32(cur[-1][1] - first_pt[1]))
            if circle and first_id == 0 and cur[-1][3] == N - 1 and float(np.sqrt(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])))) < 0.1:
                clusters[0].extend(cur)
            elif length_ok(cur):
                clusters.append(list(cur))
            if circle and cur[-1][3] == 0:
                fx, fy = f32(first_last[0] - cur[0][0]), f32(first_last[1] - cur[0][1])
                fl = np.sqrt(f32(f32(fx * fx) + f32(fy * fy)))
                if abs(float(fl) - reflector_min_length) >= reflector_length_error:
                    clusters.pop(0)
        elif length_ok(cur):
            clusters.append(list(cur))
    vidx = np.nonzero(valid)[0]
    if vidx.size == 0:
        return np.zeros((0, 2), f32), [], np.zeros((0, 2), f32)
    # ---- de-skew of every return into the last point's frame (:239-259)
    t_last = float(t_of(vidx[-1]))
    max_pose = extrap.pose(t_last)
    inv_last = _inv_d(max_pose)
    returns = np.zeros((vidx.size, 2), f32)
    for k, j in enumerate(vidx):
        rel = _cast(_mul_d(inv_last, extrap.pose(float(t_of(j)))))
        returns[k] = _apply_f(rel, *point_of_valid(j))
    to_base = _cast(_inv_d(max_pose))
    centers = []
    for c in clusters:                                          # :277-306: float32 running sum in member order
        cx = cy = f32(0)
        for x, y, t, _ in c:
            ox, oy = _apply_f(_cast(extrap.pose(float(t))), x, y)
            bx, by = _apply_f(to_base, ox, oy)
            cx, cy = f32(cx + bx), f32(cy + by)
        centers.append((f32(cx / f32(len(c))), f32(cy / f32(len(c)))))
    return (np.array(centers, f32).reshape(-1, 2), [np.array([p[3] for p in c], np.int32) for c in clusters], returns)
