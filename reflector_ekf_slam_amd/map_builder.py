"""mapping::MapBuilder of the reference (include/mapping/map_builder.h:22-73, src/mapping/map_builder.cc) on top of the
grid front-end's C ABI (include/rgrid.h): same method names, argument meaning and control flow; every per-point or
per-cell step runs on the GPU (GridFrontEnd), the host keeps what the reference's host keeps -- a handful of rigid
transforms and the decisions between the steps.

Poses are 2-D here: the reference's caller builds its Rigid3d from (x, y, yaw) (src/ros_node.cc:547-549) and projects
the result back (`transform::Project2D(match_result->local_pose)`, :468,555), so `ekf_pose` is (x, y, yaw) and
`MatchingResult.local_pose` is (x, y, yaw).  The float32 round trips the reference makes through Eigen quaternions
(Rigid3d -> cast<float> -> Project2D -> GetYaw, transform.h:27-41,93-98) are restated operation by operation below; Eigen
is not in the image, so that restatement is unpinned like the rest of the Eigen-dependent arithmetic.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from .grid import (AdaptiveVoxelFilterOptions, CeresScanMatcherOptions2D, GridFrontEnd, RangeDataInserterOptions,
                   RealTimeCorrelativeScanMatcherOptions)

_f32 = np.float32


@dataclass
class MapBuilderOptions:
    """mapping::MapBuilderOptions (map_builder.h:22-30); defaults = src/ros_node.cc:299-396."""
    resolution: float = 0.05
    voxel_filter_size: float = 0.025
    adaptive_voxel_options: AdaptiveVoxelFilterOptions = field(default_factory=AdaptiveVoxelFilterOptions)
    real_time_scan_matcher_options: RealTimeCorrelativeScanMatcherOptions = field(default_factory=RealTimeCorrelativeScanMatcherOptions)
    ceres_scan_matcher_options: CeresScanMatcherOptions2D = field(default_factory=CeresScanMatcherOptions2D)
    range_data_inserter_options: RangeDataInserterOptions = field(default_factory=RangeDataInserterOptions)


@dataclass
class RangeData:
    """sensor::RangeData (range_data.h:15-20): origin (2,), returns (n, 2), misses (m, 2), float32."""
    origin: np.ndarray
    returns: np.ndarray
    misses: np.ndarray


@dataclass
class MatchingResult:
    """mapping::MatchingResult (map_builder.h:32-37), local_pose projected to (x, y, yaw)."""
    time: float
    local_pose: np.ndarray
    range_data_in_local: RangeData


# ---- the reference's rigid-transform arithmetic, restated ----------------------------------------------------------
def yaw_of_quaternion_f32(w, z) -> np.float32:
    """transform::GetYaw(Quaternionf(w, 0, 0, z)) (transform.h:27-33): Eigen's q * UnitX = v + w * uv + vec x uv with
    uv = 2 (vec x v) gives (1 - z * 2z, w * 2z, 0); then atan2 in float."""
    w, z = _f32(w), _f32(z)
    two_z = _f32(z + z)
    return _f32(math.atan2(float(_f32(w * two_z)), float(_f32(_f32(1) - _f32(z * two_z)))))


def yaw_of_quaternion_f64(w: float, z: float) -> float:
    two_z = z + z
    return math.atan2(w * two_z, 1.0 - z * two_z)


def rigid2f_apply(translation, yaw, points) -> np.ndarray:
    """transform::Rigid2f(translation, Rotation2Df(yaw)) * p for every row of `points` (rigid_transform.h:87-93)."""
    c, s = _f32(math.cos(float(yaw))), _f32(math.sin(float(yaw)))
    p = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 2)
    t = np.asarray(translation, dtype=np.float32)
    out = np.empty_like(p)
    out[:, 0] = (c * p[:, 0] - s * p[:, 1]) + t[0]
    out[:, 1] = (s * p[:, 0] + c * p[:, 1]) + t[1]
    return out


def transform_range_data(rd: RangeData, translation, yaw) -> RangeData:
    """sensor::TransformRangeData(range_data, Rigid2f) (src/sensor/range_data.cc:7-16)."""
    return RangeData(rigid2f_apply(translation, yaw, rd.origin.reshape(1, 2))[0], rigid2f_apply(translation, yaw, rd.returns),
                     rigid2f_apply(translation, yaw, rd.misses))


class MapBuilder:
    """mapping::MapBuilder (src/mapping/map_builder.cc:6-135).  `front_end` lets several builders share one
    GridFrontEnd handle; by default the builder owns one (and with it one HIP stream and the resident grid)."""

    kInitialSubmapSize = 100                                   # map_builder.cc:115

    def __init__(self, options: MapBuilderOptions | None = None, front_end=None, max_points: int = 16384,
                 max_cells: int = 4096 * 4096):
        self.options_ = options or MapBuilderOptions()
        self._fe = front_end if front_end is not None else GridFrontEnd(max_points=max_points, max_cells=max_cells)
        self._have_submap = False
        self._submap_origin = None                              # Submap2D local_pose translation (submap_2d.cc:18-24)
        self.num_range_data = 0

    # MapBuilder::TransformToGravityAlignedFrameAndFilter  (map_builder.cc:20-32)
    def TransformToGravityAlignedFrameAndFilter(self, yaw_f32, range_data: RangeData) -> RangeData:
        cropped = transform_range_data(range_data, (0.0, 0.0), yaw_f32)
        size = self.options_.voxel_filter_size
        return RangeData(cropped.origin, self._fe.VoxelFilter(cropped.returns, size), self._fe.VoxelFilter(cropped.misses, size))

    # MapBuilder::ScanMatch  (map_builder.cc:34-55)
    def ScanMatch(self, time, pose_prediction, cloud) -> np.ndarray:
        if not self._have_submap:
            return np.array(pose_prediction, dtype=np.float64)
        coarse = self._fe.Match(pose_prediction, cloud, self.options_.real_time_scan_matcher_options)
        fine = self._fe.RefineMatch(pose_prediction[:2], coarse.pose_estimate, cloud, self.options_.ceres_scan_matcher_options)
        self.last_score = coarse.score
        self.last_summary = fine
        return fine.pose_estimate

    # MapBuilder::AddRangeData  (map_builder.cc:57-108)
    def AddRangeData(self, time: float, range_data: RangeData, ekf_pose) -> MatchingResult | None:
        rd = RangeData(np.asarray(range_data.origin, np.float32).reshape(2), np.asarray(range_data.returns, np.float32).reshape(-1, 2),
                       np.asarray(range_data.misses, np.float32).reshape(-1, 2))
        if rd.returns.shape[0] == 0:
            return None                                                          # "Dropped empty horizontal range data."
        x, y, theta = (float(v) for v in ekf_pose)
        qw, qz = math.cos(theta / 2), math.sin(theta / 2)                        # the caller's quaternion (src/ros_node.cc:548)
        # gravity_alignment = Rotation(ekf_pose.rotation()); .cast<float>() -> Project2D -> Rigid2f(0, GetYaw)
        gravity_aligned = self.TransformToGravityAlignedFrameAndFilter(yaw_of_quaternion_f32(qw, qz), rd)
        # pose_prediction = Project2D(ekf_pose * gravity_alignment.inverse()): the rotations cancel
        pose_prediction = np.array([x, y, 0.0])
        filtered = self._fe.AdaptiveVoxelFilter(gravity_aligned.returns, self.options_.adaptive_voxel_options)
        if filtered.shape[0] == 0:
            return None
        est = self.ScanMatch(time, pose_prediction, filtered)
        # pose_estimate = Embed3D(*pose_estimate_2d) * gravity_alignment: quaternion product about z, in double
        aw, az = math.cos(0.5 * est[2]), math.sin(0.5 * est[2])
        pw, pz = aw * qw - az * qz, aw * qz + az * qw
        local_pose = np.array([est[0], est[1], yaw_of_quaternion_f64(pw, pz)])
        # range_data_in_local = TransformRangeData(range_data, pose_estimate.cast<float>())
        in_local = transform_range_data(rd, (est[0], est[1]), yaw_of_quaternion_f32(pw, pz))
        # range_data_in_local2 = TransformRangeData(gravity_aligned_range_data, Embed3D(pose_estimate_2d->cast<float>()))
        af = _f32(est[2])
        ha = _f32(_f32(0.5) * af)                                                # AngleAxisf -> Quaternionf: cos / sin of the half angle
        yaw2 = yaw_of_quaternion_f32(math.cos(float(ha)), math.sin(float(ha)))
        in_local2 = transform_range_data(gravity_aligned, (est[0], est[1]), yaw2)
        self.InsertIntoSubmap(in_local2)
        return MatchingResult(time, local_pose, in_local)

    # MapBuilder::InsertIntoSubmap + CreateGrid  (map_builder.cc:110-126)
    def InsertIntoSubmap(self, range_data_in_local: RangeData) -> None:
        if not self._have_submap:
            n = self.kInitialSubmapSize
            resolution = float(_f32(self.options_.resolution))                   # `float resolution = options_.resolution`
            half = 0.5 * n * resolution
            origin = range_data_in_local.origin
            self._fe.SetGrid(np.zeros((n, n), np.uint16), resolution, (float(origin[0]) + half, float(origin[1]) + half))
            self._submap_origin = (float(origin[0]), float(origin[1]))
            self._have_submap = True
        self._fe.Insert(range_data_in_local.origin, range_data_in_local.returns, range_data_in_local.misses,
                        self.options_.range_data_inserter_options)              # GrowAsNeeded + CastRays + FinishUpdate
        self.num_range_data += 1                                                 # Submap2D::InsertRangeData (submap_2d.cc:27-35)

    # MapBuilder::ToSubmapTexture  (map_builder.cc:128-134) -> Submap2D::GetMapTextureData -> DrawToSubmapTexture
    def ToSubmapTexture(self):
        """None without a submap; else a dict with the reference's SubmapTexture fields (grid_2d.h:16-24): `cells` are the
        raw (value, alpha) bytes (the reference gzips that string), slice_pose / global_pose as (x, y) translations."""
        if not self._have_submap:
            return None
        cells, box, slice_max = self._fe.DrawTexture()
        _, _, resolution, _, _ = self._fe.GetLimits()
        ox, oy = self._submap_origin
        return {"cells": cells, "width": box[2], "height": box[3], "resolution": resolution,
                "slice_pose": (slice_max[0] - ox, slice_max[1] - oy), "global_pose": (ox, oy)}

    # introspection used by the tests
    def grid(self):
        return self._fe.GetGrid(), self._fe.GetLimits()
