"""tests/witness/detect3d_witness.py -- TEST INFRASTRUCTURE: a SECOND, independent statement of the reference's 3D
reflector detector (/root/reference/src/reflector_detect/point_cloud/point_cloud_reflector_detect.cc:9-106), written
against scipy's kd-tree instead of the C oracle's brute-force loops.  It shares no code with oracle/detect3d_oracle.c;
the two must agree (tests/test_witness_cpu.py), and its outputs are committed as fixtures (tests/golden/witness_3d.npz).

Like the oracle it restates PCL 1.7 from the published algorithms -- the reference ships no tests and PCL is absent, so this
does NOT pin parity with the reference (parity unpinned); it pins the oracle against a different implementation.

Structure (deliberately different from the oracle):
  * neighbours come from scipy.spatial.cKDTree (k-NN / ball queries in float64) with a safety margin, and are then re-ranked
    with the float32 squared distance PCL's FLANN L2_Simple computes -- the oracle scans all pairs;
  * clusters are connected components of the radius graph (scipy.sparse.csgraph) -- the oracle grows regions with a queue;
    a seeded region growing in point order visits exactly the components, in order of their smallest member.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components
from scipy.spatial import cKDTree

MEAN_K, STD_MUL, TOL, MIN_SIZE, MAX_SIZE = 30, 0.5, 0.2, 4, 160     # point_cloud_reflector_detect.cc:45,46,69-71

_libm = ctypes.CDLL("libm.so.6")
_libm.cosf.restype = ctypes.c_float
_libm.cosf.argtypes = [ctypes.c_float]
_libm.sinf.restype = ctypes.c_float
_libm.sinf.argtypes = [ctypes.c_float]


def _d2_f32(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """FLANN L2_Simple on float32 points: ((dx*dx) + dy*dy) + dz*dz, every operation rounded to float32."""
    d = (a - b).astype(np.float32)
    r = (d[..., 0] * d[..., 0]).astype(np.float32)
    r = (r + (d[..., 1] * d[..., 1]).astype(np.float32)).astype(np.float32)
    r = (r + (d[..., 2] * d[..., 2]).astype(np.float32)).astype(np.float32)
    return r


def detect3d_witness(xyzi, intensity_min=160.0, sensor_to_base_link=(0.0, 0.0, 0.0)):
    """-> (centers (K,2) float32, n_after_intensity, n_after_sor)."""
    xyzi = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
    pts = xyzi[xyzi[:, 3].astype(np.float64) > float(intensity_min), :3]          # :31-37
    M = pts.shape[0]
    # ---- StatisticalOutlierRemoval, MeanK = 30, StddevMulThresh = 0.5 (:43-47)
    dist = np.zeros(M, np.float32)
    valid = 0
    if M >= MEAN_K + 1:
        tree = cKDTree(pts.astype(np.float64))
        kk = min(M, MEAN_K + 1 + 16)                            # margin: float32 rounding may reorder near-equal neighbours
        _, nb = tree.query(pts.astype(np.float64), k=kk)
        for i in range(M):
            d2 = np.sort(_d2_f32(pts[i][None, :], pts[nb[i]]))[: MEAN_K + 1]
            s = 0.0
            for v in np.sqrt(d2[1:]):                           # sqrtf per neighbour, the sum in double, the query itself skipped
                s += float(v)
            dist[i] = np.float32(s / MEAN_K)
            valid += 1
    total, sq = 0.0, 0.0
    for v in dist:
        total += float(v)
        sq += float(v) * float(v)
    if valid > 1:
        mean = total / valid
        var = (sq - total * total / valid) / (valid - 1)
        thr = mean + STD_MUL * math.sqrt(var) if var >= 0 else float("nan")
    elif valid == 1:
        thr = float("nan")                                      # 0/0 variance: NaN threshold keeps everything
    else:
        thr = float("nan")
    keep = ~(dist.astype(np.float64) > thr)
    q = pts[keep]
    M2 = q.shape[0]
    # ---- EuclideanClusterExtraction: tolerance 0.2, sizes [4, 160] (:65-74)
    centers = np.zeros((0, 2), np.float32)
    if M2 > 0:
        tree = cKDTree(q.astype(np.float64))
        pairs = tree.query_pairs(TOL * 1.001, output_type="ndarray")
        tol2 = np.float32(TOL * TOL)
        if pairs.shape[0]:
            ok = _d2_f32(q[pairs[:, 0]], q[pairs[:, 1]]) < tol2                  # FLANN radius search: squared float32 distance
            pairs = pairs[ok]
        g = coo_matrix((np.ones(pairs.shape[0], np.int8), (pairs[:, 0], pairs[:, 1])), shape=(M2, M2))
        ncomp, lab = connected_components(g, directed=False)
        comps = [np.nonzero(lab == c)[0] for c in range(ncomp)]
        comps = [c for c in comps if MIN_SIZE <= c.size <= MAX_SIZE]
        comps.sort(key=lambda c: (-c.size, int(c[0])))          # size descending; ties (unspecified in PCL): first member ascending
        sx, sy, sa = (np.float32(v) for v in sensor_to_base_link)
        cs, sn = np.float32(_libm.cosf(float(sa))), np.float32(_libm.sinf(float(sa)))
        out = []
        for c in comps:
            cx = cy = np.float32(0.0)
            for i in c:                                         # compute3DCentroid: float32 running sum in index order
                cx = np.float32(cx + q[i, 0])
                cy = np.float32(cy + q[i, 1])
            cx = np.float32(cx / np.float32(c.size))
            cy = np.float32(cy / np.float32(c.size))
            x = np.float32(np.float32(np.float32(cs * cx) + np.float32(np.float32(-sn) * cy)) + sx)   # Rigid2f * point (:96)
            y = np.float32(np.float32(np.float32(sn * cx) + np.float32(cs * cy)) + sy)
            out.append((x, y))
        centers = np.array(out, np.float32).reshape(-1, 2)
    return centers, M, M2
