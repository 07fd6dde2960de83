"""Host-side mirror of reflector_detect::ReflectorDetectInterface
(/root/reference/include/reflector_detect/reflector_detect_interface.h:23-38) over the C ABI
of include/rdet.h.  ROS message types are replaced by plain dataclasses with the same field
names; all arithmetic happens in the HIP kernels behind librdet.so (no CPU fallback)."""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .ekf_slam import Observation, OdometryData

MAX_CENTERS = 256


class RdetError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        super().__init__(f"{where}: {_rdet().rdet_strerror(code).decode()} ({code})")


class Rdet2dOptions(C.Structure):
    _fields_ = [("intensity_min", C.c_double), ("reflector_min_length", C.c_double),
                ("reflector_length_error", C.c_double), ("range_min", C.c_float), ("range_max", C.c_float)]


class Rdet3dOptions(C.Structure):
    _fields_ = [("intensity_min", C.c_double)]


_lib_rdet = None


def _rdet():
    global _lib_rdet
    if _lib_rdet is not None:
        return _lib_rdet
    path = _lib.lib_path("librdet.so")
    if not os.path.exists(path):
        raise _lib.LibraryMissing(f"{path} not found: run `python __graft_entry__.py`; there is no CPU fallback")
    L = C.CDLL(path)
    vp, fp, dp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int)
    L.rdet_strerror.restype = C.c_char_p
    L.rdet_strerror.argtypes = [C.c_int]
    L.rdet2d_create.argtypes = [C.POINTER(Rdet2dOptions), dp, C.c_int, C.c_int, C.POINTER(vp)]
    L.rdet2d_destroy.argtypes = [vp]
    L.rdet2d_destroy.restype = None
    L.rdet2d_set_sensor_to_base_link.argtypes = [vp, dp]
    L.rdet2d_handle_odometry.argtypes = [vp, C.c_double, dp, dp, C.c_double, C.c_double, C.c_double]
    L.rdet2d_handle_scan.argtypes = [vp, C.c_double, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.c_float, vp, vp, C.c_int, vp, C.c_int, vp, vp]     # (int *K, double *t: addresses of per-handle result slots)
    L.rdet2d_get_range_data.argtypes = [vp, fp, vp, C.c_int, ip]
    if hasattr(L, "rdet3d_create"):
        L.rdet3d_create.argtypes = [C.POINTER(Rdet3dOptions), dp, C.c_int, C.c_int, C.POINTER(vp)]
        L.rdet3d_destroy.argtypes = [vp]
        L.rdet3d_destroy.restype = None
        L.rdet3d_handle_cloud.argtypes = [vp, C.c_double, vp, C.c_int, vp, C.c_int, vp, vp]
        L.rdet3d_submit.argtypes = [vp, C.c_double, vp, C.c_int, C.c_int]
        L.rdet3d_collect.argtypes = [vp, vp, C.c_int, vp, vp]
        L.rdet3d_debug_set_path.argtypes = [vp, C.c_int]
        L.rdet3d_debug_path_counts.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    _lib_rdet = L
    return L


@dataclass
class ReflectorDetectOptions:
    """reflector_detect::ReflectorDetectOptions (laser_reflector_detect.h:8-15); defaults from
    launch/slam.launch:24-26 and src/ros_node.cc:240-265."""
    intensity_min: float = 160.0
    reflector_min_length: float = 0.18
    reflector_length_error: float = 0.06
    range_min: float = 0.3
    range_max: float = 10.0


@dataclass
class PointCloudOptions:
    """reflector_detect::PointCloudOptions (point_cloud_reflector_detect.h:37-40)."""
    intensity_min: float = 160.0


@dataclass
class LaserScan:
    """The sensor_msgs::LaserScan fields HandleLaserScan reads."""
    stamp: float
    angle_min: float
    angle_max: float
    angle_increment: float
    scan_time: float
    range_min: float
    range_max: float
    ranges: np.ndarray
    intensities: np.ndarray


@dataclass
class RangeData:
    """sensor::RangeData (range_data.h:15-20)."""
    origin: np.ndarray
    returns: np.ndarray
    misses: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.float32))


def project2d(translation_xyz, quat_wxyz):
    """transform::Project2D (transform.h:93-98): (x, y, GetYaw(rotation)); GetYaw (transform.h:27-41)
    = atan2 of the rotated unit-x direction."""
    w, x, y, z = quat_wxyz
    dx = 1.0 - 2.0 * (y * y + z * z)
    dy = 2.0 * (x * y + w * z)
    return np.array([translation_xyz[0], translation_xyz[1], math.atan2(dy, dx)], dtype=np.float64)


def _as_f32(a):
    """A float32 C-contiguous ndarray as it is, anything else converted."""
    if type(a) is np.ndarray and a.dtype == np.float32 and a.flags.c_contiguous:
        return a
    return np.ascontiguousarray(a, dtype=np.float32)


def _result_slots(det):
    """Per-handle result buffers of the detectors' per-scan calls and their addresses: (centres, K, time, &centres, &K, &time)."""
    out = det.__dict__.get("_out")
    if out is None:
        c, k, t = np.zeros((MAX_CENTERS, 2), np.float32), np.zeros(1, np.int32), np.zeros(1, np.float64)
        out = det._out = (c, k, t, c.ctypes.data, k.ctypes.data, t.ctypes.data)
    return out


class LaserReflectorDetect:
    """reflector_detect::LaserReflectorDetect (laser_reflector_detect.h:17-31)."""

    def __init__(self, options: ReflectorDetectOptions, max_beams: int = 8192, device: int = 0,
                 sensor_to_base_link=(0.0, 0.0, 0.0)):
        self._L = _rdet()
        self.options = options
        o = Rdet2dOptions(options.intensity_min, options.reflector_min_length, options.reflector_length_error,
                          options.range_min, options.range_max)
        s2b = (C.c_double * 3)(*[float(v) for v in sensor_to_base_link])
        h = C.c_void_p()
        rc = self._L.rdet2d_create(C.byref(o), s2b, int(max_beams), int(device), C.byref(h))
        if rc != 0:
            raise RdetError(rc, "rdet2d_create")
        self._h = h
        self._s2b = np.array(sensor_to_base_link, dtype=np.float64)

    def close(self):
        if getattr(self, "_h", None):
            self._L.rdet2d_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def SetSensorToBaseLinkTransform(self, xyyaw):
        """Takes the transform already projected with ``project2d`` (x, y, yaw)."""
        s2b = (C.c_double * 3)(*[float(v) for v in xyyaw])
        rc = self._L.rdet2d_set_sensor_to_base_link(self._h, s2b)
        if rc != 0:
            raise RdetError(rc, "SetSensorToBaseLinkTransform")
        self._s2b = np.array(xyyaw, dtype=np.float64)

    def HandleOdometryData(self, msg: OdometryData):
        pos = (C.c_double * 2)(float(msg.position[0]), float(msg.position[1]))
        q = (C.c_double * 2)(float(msg.orientation[3]), float(msg.orientation[0]))     # (z, w)
        rc = self._L.rdet2d_handle_odometry(self._h, float(msg.time), pos, q, float(msg.linear_velocity[0]),
                                            float(msg.linear_velocity[1]), float(msg.angular_velocity[2]))
        if rc != 0:
            raise RdetError(rc, "HandleOdometryData")

    def HandleImuData(self, msg):
        return None

    def HandleLaserScan(self, msg: LaserScan) -> Observation:
        # (the per-scan path: float32 C-contiguous arrays go through by address, the result slots are the handle's -- the marshalling
        # of fresh ctypes objects and a zeroed centre array per call was 8 us of a 24 us call)
        ranges, inten = _as_f32(msg.ranges), _as_f32(msg.intensities)
        if ranges.shape != inten.shape:
            raise ValueError("ranges and intensities differ in length")
        out = _result_slots(self)
        rc = self._L.rdet2d_handle_scan(self._h, float(msg.stamp), float(msg.angle_min), float(msg.angle_max), float(msg.angle_increment),
                                        float(msg.scan_time), float(msg.range_min), float(msg.range_max), ranges.ctypes.data, inten.ctypes.data,
                                        ranges.shape[0], out[3], MAX_CENTERS, out[4], out[5])
        if rc != 0:
            raise RdetError(rc, "HandleLaserScan")
        return Observation(float(out[2][0]), out[0][: int(out[1][0])].copy())

    def GetRangeData(self) -> RangeData:
        n = C.c_int()
        origin = (C.c_float * 2)()
        self._L.rdet2d_get_range_data(self._h, origin, None, 0, C.byref(n))
        ret = np.zeros((max(n.value, 1), 2), np.float32)
        rc = self._L.rdet2d_get_range_data(self._h, origin, ret.ctypes.data_as(C.c_void_p), ret.shape[0], C.byref(n))
        if rc != 0:
            raise RdetError(rc, "GetRangeData")
        return RangeData(np.array(origin[:], dtype=np.float32), ret[: n.value].copy())


class PointCloudReflectorDetect:
    """reflector_detect::PointCloudReflectorDetect (point_cloud_reflector_detect.h:42-52)."""

    def __init__(self, options: PointCloudOptions, max_points: int = 65536, device: int = 0,
                 sensor_to_base_link=(0.0, 0.0, 0.0)):
        self._L = _rdet()
        self.options = options
        o = Rdet3dOptions(options.intensity_min)
        s2b = (C.c_double * 3)(*[float(v) for v in sensor_to_base_link])
        h = C.c_void_p()
        rc = self._L.rdet3d_create(C.byref(o), s2b, int(max_points), int(device), C.byref(h))
        if rc != 0:
            raise RdetError(rc, "rdet3d_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.rdet3d_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def HandlePointCloud(self, stamp: float, xyzi) -> Observation:
        pts = _as_f32(xyzi)
        if pts.size & 3:
            raise ValueError("points are (x, y, z, intensity) quadruples")
        out = _result_slots(self)
        rc = self._L.rdet3d_handle_cloud(self._h, float(stamp), pts.ctypes.data, pts.size >> 2, out[3], MAX_CENTERS, out[4], out[5])
        if rc != 0:
            raise RdetError(rc, "HandlePointCloud")
        return Observation(float(out[2][0]), out[0][: int(out[1][0])].copy())

    def debug_set_path(self, mode: int) -> None:
        """Test hook: which front end the next clouds get -- 0 by the previous cloud's survivor count (default), 1 the long chain (four
        launches: count, write, scatter, boxes), 2 the short one (two launches; a cloud with more survivors than it holds is sent again
        through the long chain by the collecting call)."""
        rc = self._L.rdet3d_debug_set_path(self._h, int(mode))
        if rc != 0:
            raise RdetError(rc, "debug_set_path")

    def debug_path_counts(self):
        """(clouds sent through the short front end, of those: sent again through the long chain)"""
        a, b = C.c_ulonglong(0), C.c_ulonglong(0)
        self._L.rdet3d_debug_path_counts(self._h, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def SubmitPointCloud(self, stamp: float, xyzi) -> None:
        """First half of HandlePointCloud (rdet3d_submit): the cloud goes to the device and its kernels are enqueued; returns at once.
        At most two clouds may be submitted and not collected."""
        pts = _as_f32(xyzi)
        if pts.size & 3:
            raise ValueError("points are (x, y, z, intensity) quadruples")
        rc = self._L.rdet3d_submit(self._h, float(stamp), pts.ctypes.data, pts.size >> 2, MAX_CENTERS)
        if rc != 0:
            raise RdetError(rc, "SubmitPointCloud")

    def CollectObservation(self) -> Observation:
        """Second half (rdet3d_collect): the Observation of the oldest cloud submitted and not yet collected."""
        out = _result_slots(self)
        rc = self._L.rdet3d_collect(self._h, out[3], MAX_CENTERS, out[4], out[5])
        if rc != 0:
            raise RdetError(rc, "CollectObservation")
        return Observation(float(out[2][0]), out[0][: int(out[1][0])].copy())
