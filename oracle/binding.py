"""oracle/binding.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes binding of oracle/liboracle.so (the C restatement in ekf_oracle.c,
detect2d_oracle.c, detect3d_oracle.c).  Importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc via oracle/Makefile (idempotent)."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    L.oekf_create.restype = C.c_void_p
    L.oekf_create.argtypes = [C.c_int, C.c_double, _f64p, C.c_double, C.c_double, C.c_double]
    L.oekf_destroy.argtypes = [C.c_void_p]
    L.oekf_set_mode.argtypes = [C.c_void_p, C.c_int]
    L.od_set_threads.argtypes = [C.c_int]
    L.od_get_threads.restype = C.c_int
    L.oekf_set_map.argtypes = [C.c_void_p, _f32p, _f64p, C.c_int]
    L.oekf_handle_odometry.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]
    L.oekf_handle_observation.restype = C.c_int
    L.oekf_handle_observation.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_void_p]
    L.oekf_predict_state.argtypes = [C.c_void_p, C.c_double, _f64p, C.c_void_p]
    L.oekf_get_n.restype = C.c_int
    L.oekf_get_n.argtypes = [C.c_void_p]
    L.oekf_get_time.restype = C.c_double
    L.oekf_get_time.argtypes = [C.c_void_p]
    L.oekf_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.oekf_set_state.argtypes = [C.c_void_p, C.c_double, C.c_int, _f64p, _f64p, _f64p]
    L.oekf_get_vt.argtypes = [C.c_void_p, _f64p]
    L.oekf_get_last_match.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    L.oekf_marker_ellipses.restype = C.c_int
    L.oekf_marker_ellipses.argtypes = [C.c_void_p, _f64p]
    # ---- 2D detector oracle (detect2d_oracle.c)
    L.od2_create.restype = C.c_void_p
    L.od2_create.argtypes = [C.c_double, C.c_double, C.c_double, C.c_float, C.c_float, _f64p]
    L.od2_destroy.argtypes = [C.c_void_p]
    L.od2_handle_odometry.argtypes = [C.c_void_p] + [C.c_double] * 8
    L.od2_handle_scan.restype = C.c_int
    L.od2_handle_scan.argtypes = [C.c_void_p, C.c_double, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_float, _f32p, _f32p, C.c_int, _f32p, C.c_int, C.POINTER(C.c_double)]
    L.od2_get_returns.restype = C.c_int
    L.od2_get_returns.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.od3_handle_cloud_ex.restype = C.c_int
    L.od3_handle_cloud_ex.argtypes = [C.c_double, _f64p, _f32p, C.c_int, _f32p, C.c_int,
                                      C.POINTER(C.c_int), C.POINTER(C.c_int)]
    # ---- grid-mapper front-end oracle (grid_oracle.c)
    L.ogrid_voxel_filter.restype = C.c_int
    L.ogrid_voxel_filter.argtypes = [_f32p, C.c_int, C.c_float, _f32p]
    L.ogrid_adaptive_voxel_filter.restype = C.c_int
    L.ogrid_adaptive_voxel_filter.argtypes = [_f32p, C.c_int, C.c_double, C.c_double, C.c_double, _f32p]
    L.ogrid_value_to_probability.restype = C.c_float
    L.ogrid_value_to_probability.argtypes = [C.c_uint16]
    L.ogrid_match.restype = C.c_double
    L.ogrid_match.argtypes = [C.c_void_p, _f64p, _f32p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double,
                              C.c_double, _f64p, _i32p, _i32p]
    L.ogrid_lookup_table.argtypes = [C.c_float, C.c_void_p]
    L.ogrid_insert.restype = C.c_int
    L.ogrid_insert.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, _f32p, _f32p, C.c_int,
                               _f32p, C.c_int, C.c_float, C.c_float, C.c_int]
    _lib = L
    return L


class OracleEKF:
    """Python face of the C oracle; method names follow the reference class
    (reflector_ekf_slam.h:13-64) in snake_case."""

    def __init__(self, odom_model, init_time, init_pose, lin_cov, ang_cov, obs_cov, literal=False):
        self._L = lib()
        pose = np.ascontiguousarray(init_pose, dtype=np.float64)
        self._h = self._L.oekf_create(int(odom_model), float(init_time), pose,
                                      float(lin_cov), float(ang_cov), float(obs_cov))
        self._max_obs = 0
        if literal:
            self.set_mode(True)

    def close(self):
        if self._h:
            self._L.oekf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_threads(self, n: int):
        """Threads of the structured mode's column-parallel loops (process-wide; 1 = the reference's own build).  Bit-identical results."""
        self._L.od_set_threads(int(n))

    def set_mode(self, literal: bool):
        self._L.oekf_set_mode(self._h, 1 if literal else 0)

    def set_map(self, xy, cov):
        xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
        cov = np.ascontiguousarray(cov, dtype=np.float64).reshape(-1, 4)
        self._L.oekf_set_map(self._h, xy, cov, xy.shape[0])

    def handle_odometry(self, t, vx, vy, wz):
        self._L.oekf_handle_odometry(self._h, float(t), float(vx), float(vy), float(wz))

    def handle_observation(self, t, obs, gps_pose=None):
        obs = np.ascontiguousarray(obs, dtype=np.float32).reshape(-1, 2)
        self._max_obs = max(self._max_obs, obs.shape[0])
        gp = None
        if gps_pose is not None:
            gps = np.ascontiguousarray(gps_pose, dtype=np.float64)
            gp = gps.ctypes.data_as(C.c_void_p)
        rc = self._L.oekf_handle_observation(self._h, float(t), obs.ctypes.data_as(C.c_void_p),
                                             obs.shape[0], gp)
        if rc != 0:
            raise RuntimeError(f"oracle handle_observation failed rc={rc}")

    def predict_state(self, t, full=False):
        n = self.n
        mu = np.zeros(n)
        if full:
            sig = np.zeros((n, n), order="F")
            self._L.oekf_predict_state(self._h, float(t), mu, sig.ctypes.data_as(C.c_void_p))
            return mu, sig
        self._L.oekf_predict_state(self._h, float(t), mu, None)
        return mu, None

    @property
    def n(self):
        return self._L.oekf_get_n(self._h)

    @property
    def time(self):
        return self._L.oekf_get_time(self._h)

    def mu(self):
        m = np.zeros(self.n)
        self._L.oekf_get_state(self._h, m.ctypes.data_as(C.c_void_p), None)
        return m

    def state(self):
        n = self.n
        m = np.zeros(n)
        s = np.zeros((n, n), order="F")
        self._L.oekf_get_state(self._h, m.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p))
        return m, s

    def set_state(self, t, mu, sigma, vt=(0.0, 0.0, 0.0)):
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        n = mu.shape[0]
        sig = np.asfortranarray(sigma, dtype=np.float64)
        assert sig.shape == (n, n)
        flat = np.ascontiguousarray(sig.T).reshape(-1)  # column-major bytes
        self._L.oekf_set_state(self._h, float(t), n, mu, flat, np.ascontiguousarray(vt, dtype=np.float64))

    def vt(self):
        v = np.zeros(3)
        self._L.oekf_get_vt(self._h, v)
        return v

    def marker_ellipses(self):
        """src/ros_node.cc:736-765 restated: (L,5) = mx, my, angle, x_len, y_len."""
        out = np.zeros((max((self.n - 3) // 2, 1), 5))
        k = self._L.oekf_marker_ellipses(self._h, out.reshape(-1))
        return out[:k].copy()

    def last_match(self):
        cap = max(self._max_obs, 1)
        sp = np.zeros((cap, 2), np.int32)
        mp = np.zeros((cap, 2), np.int32)
        nw = np.zeros((cap,), np.int32)
        ns, nm, nn = C.c_int(0), C.c_int(0), C.c_int(0)
        self._L.oekf_get_last_match(self._h, C.byref(ns), sp.ctypes.data_as(C.c_void_p),
                                    C.byref(nm), mp.ctypes.data_as(C.c_void_p),
                                    C.byref(nn), nw.ctypes.data_as(C.c_void_p))
        return sp[: ns.value].copy(), mp[: nm.value].copy(), nw[: nn.value].copy()


class OracleDetect2D:
    """Python face of oracle/detect2d_oracle.c (LaserReflectorDetect + PoseExtrapolator)."""

    def __init__(self, intensity_min=160.0, reflector_min_length=0.18, reflector_length_error=0.06,
                 range_min=0.3, range_max=10.0, sensor_to_base_link=(0.0, 0.0, 0.0)):
        self._L = lib()
        s2b = np.ascontiguousarray(sensor_to_base_link, dtype=np.float64)
        self._h = self._L.od2_create(intensity_min, reflector_min_length, reflector_length_error,
                                     range_min, range_max, s2b)

    def close(self):
        if self._h:
            self._L.od2_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def handle_odometry(self, t, px, py, qz, qw, vx, vy, wz):
        self._L.od2_handle_odometry(self._h, t, px, py, qz, qw, vx, vy, wz)

    def handle_scan(self, scan, max_centers=256):
        """scan: object with the LaserScan fields.  Returns (obs_time, centers[K,2]); raises on a bad scan."""
        r = np.ascontiguousarray(scan.ranges, dtype=np.float32)
        it = np.ascontiguousarray(scan.intensities, dtype=np.float32)
        out = np.zeros((max_centers, 2), np.float32)
        t = C.c_double()
        k = self._L.od2_handle_scan(self._h, float(scan.stamp), scan.angle_min, scan.angle_max, scan.angle_increment,
                                    scan.scan_time, scan.range_min, scan.range_max, r, it, r.shape[0],
                                    out.reshape(-1), max_centers, C.byref(t))
        if k < 0:
            raise ValueError(f"oracle detect2d rc={k}")
        return t.value, out[:k].copy()

    def returns(self):
        n = self._L.od2_get_returns(self._h, None, 0)
        out = np.zeros((max(n, 1), 2), np.float32)
        self._L.od2_get_returns(self._h, out.ctypes.data_as(C.c_void_p), n)
        return out[:n].copy()


def oracle_detect3d(xyzi, intensity_min=160.0, sensor_to_base_link=(0.0, 0.0, 0.0), max_centers=256):
    """oracle/detect3d_oracle.c: returns (centers[K,2], n_after_intensity, n_after_sor)."""
    L = lib()
    pts = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
    out = np.zeros((max_centers, 2), np.float32)
    m1, m2 = C.c_int(), C.c_int()
    k = L.od3_handle_cloud_ex(float(intensity_min), np.ascontiguousarray(sensor_to_base_link, dtype=np.float64),
                              pts.reshape(-1), pts.shape[0], out.reshape(-1), max_centers, C.byref(m1), C.byref(m2))
    if k < 0:
        raise ValueError(f"oracle detect3d rc={k}")
    return out[:k].copy(), m1.value, m2.value


def oracle_voxel_filter(xy, resolution):
    """sensor::VoxelFilter::Filter (voxel_filter.cc:81-95): first point of every voxel, input order."""
    L = lib()
    pts = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
    out = np.zeros((max(pts.shape[0], 1), 2), np.float32)
    m = L.ogrid_voxel_filter(pts.reshape(-1), pts.shape[0], float(resolution), out.reshape(-1))
    return out[:m].copy()


def oracle_adaptive_voxel_filter(xy, max_length=0.9, min_num_points=500, max_range=100.0):
    """sensor::AdaptiveVoxelFilter::Filter (voxel_filter.cc:116-120); defaults = src/ros_node.cc:312-322."""
    L = lib()
    pts = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
    out = np.zeros((max(pts.shape[0], 1), 2), np.float32)
    m = L.ogrid_adaptive_voxel_filter(pts.reshape(-1), pts.shape[0], float(max_length), float(min_num_points),
                                      float(max_range), out.reshape(-1))
    return out[:m].copy()


class _MatchOpt(C.Structure):
    _fields_ = [("linear_search_window", C.c_double), ("angular_search_window", C.c_double),
                ("translation_delta_cost_weight", C.c_double), ("rotation_delta_cost_weight", C.c_double)]


def oracle_match(initial_pose, points_xy, cells, resolution, max_xy, linear_search_window=0.2,
                 angular_search_window=math.radians(15.0), translation_delta_cost_weight=1e-1,
                 rotation_delta_cost_weight=1e-1):
    """RealTimeCorrelativeScanMatcher2D::Match (real_time_correlative_scan_matcher_2d.cc:84-118); option defaults =
    src/ros_node.cc:329-344.  cells: uint16 (num_y_cells, num_x_cells) correspondence-cost values.
    Returns (score, pose_estimate[3], best (scan_index, x_off, y_off), info (num_scans, num_linear, num_candidates))."""
    L = lib()
    pts = np.ascontiguousarray(points_xy, dtype=np.float32).reshape(-1, 2)
    g = np.ascontiguousarray(cells, dtype=np.uint16)
    opt = _MatchOpt(linear_search_window, angular_search_window, translation_delta_cost_weight, rotation_delta_cost_weight)
    pose = np.zeros(3)
    best = np.zeros(3, np.int32)
    info = np.zeros(3, np.int32)
    score = L.ogrid_match(C.byref(opt), np.ascontiguousarray(initial_pose, dtype=np.float64), pts.reshape(-1), pts.shape[0],
                          g.ctypes.data_as(C.c_void_p), g.shape[1], g.shape[0], float(resolution), float(max_xy[0]),
                          float(max_xy[1]), pose, best, info)
    return score, pose, tuple(int(v) for v in best), tuple(int(v) for v in info)


def oracle_lookup_table(probability):
    """ComputeLookupTableToApplyCorrespondenceCostOdds(Odds(probability)) (probability_values.cc:76-96): uint16[32768]."""
    t = np.zeros(32768, np.uint16)
    lib().ogrid_lookup_table(float(probability), t.ctypes.data_as(C.c_void_p))
    return t


def oracle_insert(cells, resolution, max_xy, origin, returns_xy, misses_xy=None, hit_probability=0.55,
                  miss_probability=0.49, insert_free_space=True):
    """ProbabilityGridRangeDataInserter2D::Insert on a copy of `cells` (no growth: everything must be inside);
    option defaults = src/ros_node.cc:386-396.  Returns the updated uint16 grid."""
    g = np.array(cells, dtype=np.uint16, order="C", copy=True)
    ret = np.ascontiguousarray(returns_xy, dtype=np.float32).reshape(-1, 2)
    mis = np.zeros((0, 2), np.float32) if misses_xy is None else np.ascontiguousarray(misses_xy, dtype=np.float32).reshape(-1, 2)
    rc = lib().ogrid_insert(g.ctypes.data_as(C.c_void_p), g.shape[1], g.shape[0], float(resolution), float(max_xy[0]),
                            float(max_xy[1]), np.ascontiguousarray(origin, dtype=np.float32), ret.reshape(-1) if ret.size else np.zeros(1, np.float32),
                            ret.shape[0], mis.reshape(-1) if mis.size else np.zeros(1, np.float32), mis.shape[0],
                            float(hit_probability), float(miss_probability), 1 if insert_free_space else 0)
    if rc != 0:
        raise ValueError("oracle insert: a point lies outside the grid (the caller must grow it first)")
    return g


def oracle_grow(cells, resolution, max_xy, origin, returns_xy, misses_xy=None):
    """GrowAsNeeded + Grid2D::GrowLimits (probability_grid_range_data_inserter_2d.cc:20-38, grid_2d.cc:59-99):
    returns (grown uint16 grid, new max_xy, offset of the old cell (0, 0))."""
    g = np.ascontiguousarray(cells, dtype=np.uint16)
    ret = np.ascontiguousarray(returns_xy, dtype=np.float32).reshape(-1, 2)
    mis = np.zeros((0, 2), np.float32) if misses_xy is None else np.ascontiguousarray(misses_xy, dtype=np.float32).reshape(-1, 2)
    dims = (C.c_int * 2)(g.shape[1], g.shape[0])
    mx = (C.c_double * 2)(float(max_xy[0]), float(max_xy[1]))
    off = (C.c_int * 2)()
    L = lib()
    L.ogrid_grow_limits.restype = None
    L.ogrid_grow_limits.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    org = np.ascontiguousarray(origin, dtype=np.float32)
    r0 = ret.reshape(-1) if ret.size else np.zeros(1, np.float32)
    m0 = mis.reshape(-1) if mis.size else np.zeros(1, np.float32)
    L.ogrid_grow_limits(dims, float(resolution), mx, org.ctypes.data_as(C.c_void_p), r0.ctypes.data_as(C.c_void_p), ret.shape[0],
                        m0.ctypes.data_as(C.c_void_p), mis.shape[0], off)
    new = np.zeros((dims[1], dims[0]), np.uint16)
    L.ogrid_grow_copy.restype = None
    L.ogrid_grow_copy.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.ogrid_grow_copy(g.ctypes.data_as(C.c_void_p), g.shape[1], g.shape[0], new.ctypes.data_as(C.c_void_p), dims[0], dims[1], off)
    return new, (mx[0], mx[1]), (off[0], off[1])


def oracle_refine_match(target_translation, initial_pose, points_xy, cells, resolution, max_xy, occupied_space_weight=1.0,
                        translation_weight=0.1, rotation_weight=0.4, max_num_iterations=100, use_nonmonotonic_steps=True):
    """CeresScanMatcher2D::Match (ceres_scan_matcher_2d.cc:26-62; option defaults = src/ros_node.cc:350-377).
    Returns (pose_estimate[3], summary dict)."""
    g = np.ascontiguousarray(cells, dtype=np.uint16)
    pts = np.ascontiguousarray(points_xy, dtype=np.float32).reshape(-1, 2)
    L = lib()
    L.ogrid_refine_match.restype = C.c_int
    L.ogrid_refine_match.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double,
                                     C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    opt = np.array([occupied_space_weight, translation_weight, rotation_weight, max_num_iterations, 1.0 if use_nonmonotonic_steps else 0.0])
    tt = np.array(target_translation, dtype=np.float64)
    ip = np.array(initial_pose, dtype=np.float64)
    pose = np.zeros(3)
    summ = np.zeros(4)
    rc = L.ogrid_refine_match(opt.ctypes.data_as(C.c_void_p), tt.ctypes.data_as(C.c_void_p), ip.ctypes.data_as(C.c_void_p),
                              pts.ctypes.data_as(C.c_void_p), pts.shape[0], g.ctypes.data_as(C.c_void_p), g.shape[1], g.shape[0],
                              float(resolution), float(max_xy[0]), float(max_xy[1]), pose.ctypes.data_as(C.c_void_p),
                              summ.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError("oracle refine: empty point cloud")
    return pose, {"initial_cost": summ[0], "final_cost": summ[1], "iterations": int(summ[2]), "termination": int(summ[3])}


def oracle_draw_texture(cells, resolution, max_xy):
    """ProbabilityGrid::DrawToSubmapTexture (probability_grid.cc:86-131) without gzip:
    returns (uint8 array (height, width, 2), box (offset_x, offset_y, width, height), slice_max (x, y))."""
    g = np.ascontiguousarray(cells, dtype=np.uint16)
    out = np.zeros(2 * g.size + 2, np.uint8)
    box = (C.c_int * 4)()
    sm = (C.c_double * 2)()
    L = lib()
    L.ogrid_draw_texture.restype = C.c_long
    L.ogrid_draw_texture.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
    nb = L.ogrid_draw_texture(g.ctypes.data_as(C.c_void_p), g.shape[1], g.shape[0], float(resolution), float(max_xy[0]), float(max_xy[1]),
                              out.ctypes.data_as(C.c_void_p), out.size, box, sm)
    assert nb == 2 * box[2] * box[3]
    return out[:nb].reshape(box[3], box[2], 2).copy(), tuple(box[:]), (sm[0], sm[1])
