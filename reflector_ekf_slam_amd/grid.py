"""ctypes mirror of the grid-mapper front-end (include/rgrid.h): sensor::VoxelFilter, sensor::AdaptiveVoxelFilter
(src/sensor/voxel_filter.cc) and scan_matching::RealTimeCorrelativeScanMatcher2D (src/scan_matching/
real_time_correlative_scan_matcher_2d.cc) of the reference, on the device.  No CPU fallback."""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass

import numpy as np

from . import _lib


class RgridError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where}: rgrid error {code}" + (f" ({detail})" if detail else ""))


class _MatchOptions(C.Structure):
    _fields_ = [("linear_search_window", C.c_double), ("angular_search_window", C.c_double),
                ("translation_delta_cost_weight", C.c_double), ("rotation_delta_cost_weight", C.c_double)]


_rgrid = None
RGRID_ABI_VERSION = 3            # include/rgrid.h


def _lib_rgrid():
    global _rgrid
    if _rgrid is not None:
        return _rgrid
    path = _lib.lib_path("librgrid.so")
    if not os.path.exists(path):
        raise _lib.LibraryMissing(f"{path} not found: the HIP extension is not built "
                                  "(run `python __graft_entry__.py`); there is no CPU fallback")
    L = C.CDLL(path)
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)
    L.rgrid_abi_version.restype = C.c_int
    if L.rgrid_abi_version() != RGRID_ABI_VERSION:
        raise _lib.LibraryMissing(f"{path} has ABI version {L.rgrid_abi_version()}, this package expects {RGRID_ABI_VERSION}: "
                                  "rebuild it (`python __graft_entry__.py`)")
    L.rgrid_strerror.restype = C.c_char_p
    L.rgrid_strerror.argtypes = [C.c_int]
    L.rgrid_last_hip_error.restype = C.c_char_p
    L.rgrid_last_hip_error.argtypes = [vp]
    L.rgrid_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.rgrid_destroy.argtypes = [vp]
    L.rgrid_destroy.restype = None
    L.rgrid_voxel_filter.argtypes = [vp, vp, C.c_int, C.c_float, vp, C.c_int, ip]
    L.rgrid_adaptive_voxel_filter.argtypes = [vp, vp, C.c_int, C.c_double, C.c_double, C.c_double, vp, C.c_int, ip]
    L.rgrid_set_grid.argtypes = [vp, vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
    L.rgrid_match.argtypes = [vp, C.POINTER(_MatchOptions), dp, vp, C.c_int, dp, dp, ip, ip]
    L.rgrid_insert.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, C.c_float, C.c_float, C.c_int]
    L.rgrid_get_grid.argtypes = [vp, vp, C.c_long]
    L.rgrid_grow_as_needed.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int]
    L.rgrid_get_limits.argtypes = [vp, ip, ip, dp, dp, dp]
    L.rgrid_draw_texture.argtypes = [vp, vp, C.c_long, ip, dp]
    L.rgrid_add_range_data.argtypes = [vp, C.POINTER(_MapBuilderOptionsC), vp, vp, C.c_int, vp, C.c_int, dp, dp, vp, ip]
    L.rgrid_refine_match.argtypes = [vp, C.POINTER(_RefineOptions), dp, dp, vp, C.c_int, dp, C.POINTER(_RefineSummary)]
    _rgrid = L
    return L


class _RefineOptions(C.Structure):
    _fields_ = [("occupied_space_weight", C.c_double), ("translation_weight", C.c_double), ("rotation_weight", C.c_double),
                ("max_num_iterations", C.c_int), ("use_nonmonotonic_steps", C.c_int)]


class _MapBuilderOptionsC(C.Structure):
    _fields_ = [("resolution", C.c_float), ("voxel_filter_size", C.c_float), ("adaptive_max_length", C.c_double),
                ("adaptive_min_num_points", C.c_double), ("adaptive_max_range", C.c_double), ("match", _MatchOptions),
                ("refine", _RefineOptions), ("hit_probability", C.c_float), ("miss_probability", C.c_float), ("insert_free_space", C.c_int)]


class _RefineSummary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("iterations", C.c_int), ("termination", C.c_int)]


@dataclass
class CeresScanMatcherOptions2D:
    """scan_matching::CeresScanMatcherOptions2D (ceres_scan_matcher_2d.h:16-22) with the two ceres::Solver::Options
    fields the reference sets; defaults = src/ros_node.cc:350-377."""
    occupied_space_weight: float = 1.0
    translation_weight: float = 0.1
    rotation_weight: float = 0.4
    max_num_iterations: int = 100
    use_nonmonotonic_steps: bool = True


@dataclass
class RefineResult:
    pose_estimate: np.ndarray
    initial_cost: float
    final_cost: float
    iterations: int
    termination: int            # 0 CONVERGENCE, 1 NO_CONVERGENCE, 2 FAILURE


@dataclass
class RealTimeCorrelativeScanMatcherOptions:
    """scan_matching::RealTimeCorrelativeScanMatcherOptions (real_time_correlative_scan_matcher_2d.h:34-40);
    defaults = the reference's caller, src/ros_node.cc:329-344 (the angular window converted to radians there)."""
    linear_search_window: float = 0.2
    angular_search_window: float = math.radians(15.0)
    translation_delta_cost_weight: float = 1e-1
    rotation_delta_cost_weight: float = 1e-1


@dataclass
class AdaptiveVoxelFilterOptions:
    """sensor::AdaptiveVoxelFilterOptions (voxel_filter.h); defaults = src/ros_node.cc:312-322."""
    max_length: float = 0.9
    min_num_points: float = 500
    max_range: float = 100.0


@dataclass
class RangeDataInserterOptions:
    """mapping::ProbabilityGridRangeDataInserterOptions2D (probability_grid_range_data_inserter_2d.h:16-21);
    defaults = src/ros_node.cc:386-396."""
    insert_free_space: bool = True
    hit_probability: float = 0.55
    miss_probability: float = 0.49


@dataclass
class MatchResult:
    score: float
    pose_estimate: np.ndarray        # x, y, angle
    best: tuple                      # (scan_index, x_index_offset, y_index_offset)
    info: tuple                      # (num_scans, num_linear_perturbations, num_candidates)


class GridFrontEnd:
    """One handle of include/rgrid.h: voxel filters + the real-time correlative scan matcher."""

    def __init__(self, max_points: int = 8192, max_cells: int = 4096 * 4096, max_candidates: int = 1 << 20, device: int = 0):
        self._L = _lib_rgrid()
        h = C.c_void_p()
        rc = self._L.rgrid_create(int(max_points), int(max_cells), int(max_candidates), int(device), C.byref(h))
        if rc != 0:
            raise RgridError(rc, "rgrid_create")
        self._h = h
        self.max_points = int(max_points)

    def close(self):
        if getattr(self, "_h", None):
            self._L.rgrid_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, where):
        if rc != 0:
            raise RgridError(rc, where, self._L.rgrid_last_hip_error(self._h).decode() if rc == -2 else
                             self._L.rgrid_strerror(rc).decode())

    # sensor::VoxelFilter(size).Filter(point_cloud)  (voxel_filter.cc:81-95)
    def VoxelFilter(self, points_xy, size: float) -> np.ndarray:
        pts = np.ascontiguousarray(points_xy, dtype=np.float32).reshape(-1, 2)
        out = np.zeros((max(pts.shape[0], 1), 2), np.float32)
        m = C.c_int()
        self._chk(self._L.rgrid_voxel_filter(self._h, pts.ctypes.data_as(C.c_void_p), pts.shape[0], float(size),
                                             out.ctypes.data_as(C.c_void_p), out.shape[0], C.byref(m)), "VoxelFilter")
        return out[: m.value].copy()

    # sensor::AdaptiveVoxelFilter(options).Filter(point_cloud)  (voxel_filter.cc:116-120)
    def AdaptiveVoxelFilter(self, points_xy, options: AdaptiveVoxelFilterOptions | None = None) -> np.ndarray:
        o = options or AdaptiveVoxelFilterOptions()
        pts = np.ascontiguousarray(points_xy, dtype=np.float32).reshape(-1, 2)
        out = np.zeros((max(pts.shape[0], 1), 2), np.float32)
        m = C.c_int()
        self._chk(self._L.rgrid_adaptive_voxel_filter(self._h, pts.ctypes.data_as(C.c_void_p), pts.shape[0],
                                                      float(o.max_length), float(o.min_num_points), float(o.max_range),
                                                      out.ctypes.data_as(C.c_void_p), out.shape[0], C.byref(m)),
                  "AdaptiveVoxelFilter")
        return out[: m.value].copy()

    def SetGrid(self, cells, resolution: float, max_xy):
        """cells: uint16 (num_y_cells, num_x_cells) correspondence-cost values (grid_2d.h:83-106); MapLimits
        resolution and max corner (map_limits.h:24-45)."""
        g = np.ascontiguousarray(cells, dtype=np.uint16)
        assert g.ndim == 2
        self._grid_shape = g.shape
        self._chk(self._L.rgrid_set_grid(self._h, g.ctypes.data_as(C.c_void_p), g.shape[1], g.shape[0], float(resolution),
                                         float(max_xy[0]), float(max_xy[1])), "SetGrid")

    # GrowAsNeeded  (probability_grid_range_data_inserter_2d.cc:20-38 -> Grid2D::GrowLimits, grid_2d.cc:59-99)
    def GrowAsNeeded(self, origin_xy, returns_xy, misses_xy=None):
        org = (C.c_float * 2)(float(origin_xy[0]), float(origin_xy[1]))
        ret = np.ascontiguousarray(returns_xy, dtype=np.float32).reshape(-1, 2)
        mis = np.zeros((0, 2), np.float32) if misses_xy is None else np.ascontiguousarray(misses_xy, dtype=np.float32).reshape(-1, 2)
        self._chk(self._L.rgrid_grow_as_needed(self._h, org, ret.ctypes.data_as(C.c_void_p) if ret.size else None, ret.shape[0],
                                               mis.ctypes.data_as(C.c_void_p) if mis.size else None, mis.shape[0]), "GrowAsNeeded")
        nx, ny, _, _, _ = self.GetLimits()
        self._grid_shape = (ny, nx)

    def GetLimits(self):
        """MapLimits of the resident grid: (num_x_cells, num_y_cells, resolution, max_x, max_y)."""
        nx, ny = C.c_int(), C.c_int()
        res, mx, my = C.c_double(), C.c_double(), C.c_double()
        self._chk(self._L.rgrid_get_limits(self._h, C.byref(nx), C.byref(ny), C.byref(res), C.byref(mx), C.byref(my)), "GetLimits")
        return nx.value, ny.value, res.value, mx.value, my.value

    # ProbabilityGridRangeDataInserter2D::Insert  (probability_grid_range_data_inserter_2d.cc:103-114); grow=True runs
    # GrowAsNeeded first, as the reference's CastRays does (:45)
    def Insert(self, origin_xy, returns_xy, misses_xy=None, options: "RangeDataInserterOptions | None" = None, grow: bool = True):
        o = options or RangeDataInserterOptions()
        if grow:
            self.GrowAsNeeded(origin_xy, returns_xy, misses_xy)
        org = (C.c_float * 2)(float(origin_xy[0]), float(origin_xy[1]))
        ret = np.ascontiguousarray(returns_xy, dtype=np.float32).reshape(-1, 2)
        mis = np.zeros((0, 2), np.float32) if misses_xy is None else np.ascontiguousarray(misses_xy, dtype=np.float32).reshape(-1, 2)
        self._chk(self._L.rgrid_insert(self._h, org, ret.ctypes.data_as(C.c_void_p) if ret.size else None, ret.shape[0],
                                       mis.ctypes.data_as(C.c_void_p) if mis.size else None, mis.shape[0],
                                       float(o.hit_probability), float(o.miss_probability), 1 if o.insert_free_space else 0),
                  "Insert")

    # CeresScanMatcher2D::Match  (ceres_scan_matcher_2d.cc:26-62)
    def RefineMatch(self, target_translation, initial_pose_estimate, point_cloud,
                    options: "CeresScanMatcherOptions2D | None" = None) -> "RefineResult":
        o = options or CeresScanMatcherOptions2D()
        co = _RefineOptions(o.occupied_space_weight, o.translation_weight, o.rotation_weight, int(o.max_num_iterations),
                            1 if o.use_nonmonotonic_steps else 0)
        pts = np.ascontiguousarray(point_cloud, dtype=np.float32).reshape(-1, 2)
        tt = (C.c_double * 2)(float(target_translation[0]), float(target_translation[1]))
        ip = (C.c_double * 3)(*[float(v) for v in initial_pose_estimate])
        pe = (C.c_double * 3)()
        sm = _RefineSummary()
        self._chk(self._L.rgrid_refine_match(self._h, C.byref(co), tt, ip, pts.ctypes.data_as(C.c_void_p), pts.shape[0], pe,
                                             C.byref(sm)), "RefineMatch")
        return RefineResult(np.array(pe[:]), sm.initial_cost, sm.final_cost, sm.iterations, sm.termination)

    # mapping::MapBuilder::AddRangeData  (map_builder.cc:57-108) as ONE C call (rgrid_add_range_data); `options` is a
    # map_builder.MapBuilderOptions.  Returns (status, local_pose (x, y, yaw), returns in the local frame): status 0 =
    # inserted, 1 = no returns, 2 = nothing left after the filters (the reference returns nullptr for 1 and 2).
    def AddRangeData(self, options, origin_xy, returns_xy, misses_xy, ekf_pose):
        o = options
        m, r = o.real_time_scan_matcher_options, o.ceres_scan_matcher_options
        a, ins = o.adaptive_voxel_options, o.range_data_inserter_options
        co = _MapBuilderOptionsC(o.resolution, o.voxel_filter_size, a.max_length, a.min_num_points, a.max_range,
                                 _MatchOptions(m.linear_search_window, m.angular_search_window, m.translation_delta_cost_weight,
                                               m.rotation_delta_cost_weight),
                                 _RefineOptions(r.occupied_space_weight, r.translation_weight, r.rotation_weight, int(r.max_num_iterations),
                                                1 if r.use_nonmonotonic_steps else 0),
                                 ins.hit_probability, ins.miss_probability, 1 if ins.insert_free_space else 0)
        org = (C.c_float * 2)(float(origin_xy[0]), float(origin_xy[1]))
        ret = np.ascontiguousarray(returns_xy, dtype=np.float32).reshape(-1, 2)
        mis = np.zeros((0, 2), np.float32) if misses_xy is None else np.ascontiguousarray(misses_xy, dtype=np.float32).reshape(-1, 2)
        pose = (C.c_double * 3)(*[float(v) for v in ekf_pose])
        out_pose = (C.c_double * 3)()
        in_local = np.zeros_like(ret)
        status = C.c_int()
        self._chk(self._L.rgrid_add_range_data(self._h, C.byref(co), org, ret.ctypes.data_as(C.c_void_p) if ret.size else None, ret.shape[0],
                                               mis.ctypes.data_as(C.c_void_p) if mis.size else None, mis.shape[0], pose, out_pose,
                                               in_local.ctypes.data_as(C.c_void_p) if ret.size else None, C.byref(status)), "AddRangeData")
        if status.value == 0:
            nx, ny, _, _, _ = self.GetLimits()
            self._grid_shape = (ny, nx)
        return status.value, np.array(out_pose[:]), in_local

    # ProbabilityGrid::DrawToSubmapTexture  (probability_grid.cc:86-131), without the gzip container
    def DrawTexture(self):
        """Returns (uint8 (height, width, 2) = (value, alpha) per cell of the known-cells window,
        box (offset_x, offset_y, width, height), slice_max (x, y))."""
        ny, nx = self._grid_shape
        out = np.zeros(2 * nx * ny, np.uint8)
        box = (C.c_int * 4)()
        sm = (C.c_double * 2)()
        self._chk(self._L.rgrid_draw_texture(self._h, out.ctypes.data_as(C.c_void_p), out.size, box, sm), "DrawTexture")
        return out[: 2 * box[2] * box[3]].reshape(box[3], box[2], 2).copy(), tuple(box[:]), (sm[0], sm[1])

    def GetGrid(self) -> np.ndarray:
        out = np.zeros(self._grid_shape, np.uint16)
        self._chk(self._L.rgrid_get_grid(self._h, out.ctypes.data_as(C.c_void_p), out.size), "GetGrid")
        return out

    # RealTimeCorrelativeScanMatcher2D::Match  (real_time_correlative_scan_matcher_2d.cc:84-118)
    def Match(self, initial_pose_estimate, point_cloud, options: RealTimeCorrelativeScanMatcherOptions | None = None) -> MatchResult:
        o = options or RealTimeCorrelativeScanMatcherOptions()
        co = _MatchOptions(o.linear_search_window, o.angular_search_window, o.translation_delta_cost_weight,
                           o.rotation_delta_cost_weight)
        pts = np.ascontiguousarray(point_cloud, dtype=np.float32).reshape(-1, 2)
        ip = (C.c_double * 3)(*[float(v) for v in initial_pose_estimate])
        pe = (C.c_double * 3)()
        sc = C.c_double()
        best = (C.c_int * 3)()
        info = (C.c_int * 3)()
        self._chk(self._L.rgrid_match(self._h, C.byref(co), ip, pts.ctypes.data_as(C.c_void_p), pts.shape[0], pe,
                                      C.byref(sc), best, info), "Match")
        return MatchResult(sc.value, np.array(pe[:]), tuple(best[:]), tuple(info[:]))
