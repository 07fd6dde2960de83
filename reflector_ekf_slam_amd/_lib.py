"""ctypes loader for the C-ABI libraries built from csrc/ (include/rekf.h, include/rdet.h).

There is deliberately NO fallback: if the HIP extension is missing or does not
load, importing the product path raises.  Build with ``python __graft_entry__.py``
(or ``make -C reflector_ekf_slam_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_OBS = 256
REKF_ABI_VERSION = 7          # must equal REKF_ABI_VERSION of include/rekf.h and rekf_abi_version() of the built library


class RekfOptions(C.Structure):
    """struct rekf_options (include/rekf.h) == ekf::EKFOptions (ekf_slam_interface.h:28-41)."""
    _fields_ = [("odom_model", C.c_int), ("use_imu", C.c_int), ("init_time", C.c_double),
                ("init_pose", C.c_double * 3), ("linear_velocity_cov", C.c_double),
                ("angular_velocity_cov", C.c_double), ("observation_cov", C.c_double)]


class LibraryMissing(RuntimeError):
    pass


def lib_path(name: str) -> str:
    return os.path.join(_HERE, name)


_rekf = None


def rekf():
    """librekf.so with argtypes set.  Raises LibraryMissing when it was not built."""
    global _rekf
    if _rekf is not None:
        return _rekf
    path = lib_path("librekf.so")
    if not os.path.exists(path):
        raise LibraryMissing(f"{path} not found: the HIP extension is not built "
                             "(run `python __graft_entry__.py`); there is no CPU fallback")
    L = C.CDLL(path)
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)
    L.rekf_abi_version.restype = C.c_int
    have = L.rekf_abi_version()
    if have != REKF_ABI_VERSION:
        raise LibraryMissing(f"{path} has ABI version {have}, this package speaks {REKF_ABI_VERSION}: rebuild it "
                             "(python __graft_entry__.py); there is no CPU fallback")
    L.rekf_strerror.restype = C.c_char_p
    L.rekf_strerror.argtypes = [C.c_int]
    L.rekf_last_hip_error.restype = C.c_char_p
    L.rekf_last_hip_error.argtypes = [vp]
    L.rekf_create.argtypes = [C.POINTER(RekfOptions), C.c_int, C.c_int, C.POINTER(vp)]
    L.rekf_destroy.argtypes = [vp]
    L.rekf_destroy.restype = None
    L.rekf_set_map.argtypes = [vp, vp, vp, C.c_int]
    L.rekf_handle_odometry.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.c_double]
    L.rekf_handle_observation.argtypes = [vp, C.c_double, vp, C.c_int, vp]
    L.rekf_predict_state.argtypes = [vp, C.c_double, dp, dp]
    L.rekf_predict_state_full.argtypes = [vp, C.c_double, dp, ip, vp, C.c_long, vp, C.c_long]
    L.rekf_get_flags.argtypes = [vp, ip]
    L.rekf_get_time.argtypes = [vp, dp]
    L.rekf_get_pose.argtypes = [vp, vp, vp, vp]          # (double *: plain addresses of a preallocated buffer, ekf_slam.py -- the per-call path)
    L.rekf_get_n.argtypes = [vp, ip]
    L.rekf_get_marker_ellipses.argtypes = [vp, vp, C.c_int, ip]
    L.rekf_get_state.argtypes = [vp, dp, ip, vp, C.c_long, vp, C.c_long]
    L.rekf_set_state.argtypes = [vp, C.c_double, C.c_int, vp, vp, vp]
    L.rekf_get_last_match.argtypes = [vp, ip, vp, ip, vp, ip, vp]
    L.rekf_sync.argtypes = [vp]
    L.rekf_profile_enable.argtypes = [vp, C.c_int]
    L.rekf_profile_read.argtypes = [vp, C.c_int, dp, C.POINTER(C.c_long)]
    L.rekf_profile_reset.argtypes = [vp]
    L.rekf_profile_samples.argtypes = [vp, vp, C.c_long, C.POINTER(C.c_long)]
    L.rekf_stream.restype = vp
    L.rekf_stream.argtypes = [vp]
    L.rekf_device_layout.argtypes = [vp, ip, ip, C.POINTER(vp), C.POINTER(vp)]
    L.rekf_reserve.argtypes = [vp, C.c_int]
    L.rekf_set_auto_grow.argtypes = [vp, C.c_int]
    L.rekf_set_exclusive.argtypes = [vp, C.c_int]
    L.rekf_get_capacity.argtypes = [vp, ip]
    L.rekf_debug_time_kernel.argtypes = [vp, C.c_int, C.c_int, C.c_int, dp]
    if hasattr(L, "rekf_debug_inject_failure"):           # (absent from older builds that scripts/gpu_ab.py compares against)
        L.rekf_debug_inject_failure.argtypes = [vp, C.c_int]
    if hasattr(L, "rekf_debug_set_grid"):
        L.rekf_debug_set_grid.argtypes = [vp, C.c_int, C.c_double, C.c_int]
    _rekf = L
    return L
