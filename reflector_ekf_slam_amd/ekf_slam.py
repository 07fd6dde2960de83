"""Host-side mirror of the reference's EKF interface over the C ABI (include/rekf.h).

``ReflectorEKFSLAM`` keeps the method names of ekf::ReflectorEKFSLAMInterface
(/root/reference/include/reflector_ekf_slam/ekf_slam_interface.h:50-67), including
the ``GetCoviarance`` spelling, so that tests read like calls into the reference.
All arithmetic happens in the HIP kernels behind librekf.so; this file only
marshals arguments.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib

DIFF, OMNI = 0, 1   # sensor::OdometryModel (sensor_data.h:56-60)

KERNELS = {"predict": 0, "front": 1, "downdate": 5, "augment": 6, "empty": 7,
           "update": 8, "mid": 9}


class RekfError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        msg = _lib.rekf().rekf_strerror(code).decode()
        super().__init__(f"{where}: {msg} ({code}) {detail}")


@dataclass
class EKFOptions:
    """ekf::EKFOptions (ekf_slam_interface.h:28-41).  *_cov are variances
    (the reference's caller squares the launch sigmas, src/ros_node.cc:207-238)."""
    use_imu: bool = False
    init_time: float = 0.0
    init_pose: tuple = (0.0, 0.0, 0.0)
    map_path: str = ""
    odom_model: int = DIFF
    linear_velocity_cov: float = 0.05 * 0.05
    angular_velocity_cov: float = 0.08 * 0.08
    observation_cov: float = 0.05 * 0.05


@dataclass
class State:
    """ekf::State (ekf_slam_interface.h:43-48)."""
    time: float
    mu: np.ndarray
    sigma: np.ndarray


@dataclass
class OdometryData:
    """sensor::OdometryData (sensor_data.h:39-46); only time, linear_velocity.x/y and
    angular_velocity.z reach the EKF (reflector_ekf_slam.cc:216)."""
    time: float
    linear_velocity: tuple = (0.0, 0.0, 0.0)
    angular_velocity: tuple = (0.0, 0.0, 0.0)
    position: tuple = (0.0, 0.0, 0.0)
    orientation: tuple = (1.0, 0.0, 0.0, 0.0)   # w, x, y, z


@dataclass
class Observation:
    """sensor::Observation (sensor_data.h:20-28)."""
    time_: float
    cloud_: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.float32))
    gps_pose_: tuple | None = None   # (x, y, yaw) -- USE_GPS build only


@dataclass
class ReflectorMatchResult:
    """ekf::ReflectorMatchResult (ekf_slam_interface.h:18-26)."""
    map_obs_match_ids: np.ndarray
    state_obs_match_ids: np.ndarray
    new_ids: np.ndarray


@dataclass
class Map:
    """sensor::Map (sensor_data.h:30-37)."""
    reflector_map_: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.float32))
    reflector_map_coviarance_: np.ndarray = field(default_factory=lambda: np.zeros((0, 2, 2)))


def load_map_txt(path: str) -> Map:
    """The two-line txt map (reflector_ekf_slam.cc:43-95) read as its author meant it:
    line 0 = x,y,...  line 1 = c00,c01,c10,c11,...  (the reference indexes line 0 for the
    covariances, :90 -- undefined behaviour we do not reproduce)."""
    import os
    if not path or not os.path.exists(path):
        return Map()
    rows = []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line:
                rows.append([float(p) for p in line.split(",") if p != ""])
    if len(rows) != 2 or len(rows[1]) != 2 * len(rows[0]):
        return Map()
    xy = np.asarray(rows[0], dtype=np.float64).reshape(-1, 2).astype(np.float32)
    cov = np.asarray(rows[1], dtype=np.float64).reshape(-1, 2, 2)
    return Map(xy, cov)


def save_map_txt(path: str, state: State, loaded: Map | None = None, reference_bytes: bool = False) -> None:
    """SaveReflectorResult (src/ros_node.cc:75-140): pre-loaded map points first, then the state's landmarks with
    their 2x2 blocks, two comma-separated lines.

    Default: lossless (%.17g) and without the leading-comma bug (Q10).  ``reference_bytes=True`` writes the very
    bytes the reference's ``std::ofstream <<`` would: 6 significant digits (iostream default = %g), map points
    formatted from their float32 values, and the "," the reference emits in front of the new landmarks even when
    no map was pre-loaded (:97-100,:124-127) -- which its own loader then chokes on; ours skips empty fields."""
    pts, covs = [], []
    n_loaded = 0
    if loaded is not None:
        for p, c in zip(loaded.reflector_map_, loaded.reflector_map_coviarance_):
            pts.append((float(np.float32(p[0])), float(np.float32(p[1]))))
            covs.append(np.asarray(c, dtype=np.float64).reshape(4))
            n_loaded += 1
    L = (state.mu.shape[0] - 3) // 2
    for j in range(L):
        pts.append((state.mu[3 + 2 * j], state.mu[4 + 2 * j]))
        covs.append(state.sigma[3 + 2 * j: 5 + 2 * j, 3 + 2 * j: 5 + 2 * j].reshape(4))
    if not reference_bytes:
        with open(path, "w") as f:
            f.write(",".join(f"{v:.17g}" for p in pts for v in p) + "\n")
            f.write(",".join(f"{v:.17g}" for c in covs for v in c) + "\n")
        return

    def line(groups):
        old = ",".join("%g" % v for g in groups[:n_loaded] for v in g)
        new = ",".join("%g" % v for g in groups[n_loaded:] for v in g)
        return old + (("," + new) if L > 0 else "")
    with open(path, "w") as f:
        f.write(line(pts) + "\n")
        f.write(line(covs) + "\n")


class ReflectorEKFSLAM:
    """ekf::ReflectorEKFSLAM (reflector_ekf_slam.h:13-64) on one MI355X."""

    def __init__(self, options: EKFOptions, max_landmarks: int = 1024, device: int = 0, auto_grow: bool = True):
        self._L = _lib.rekf()
        self.options = options
        o = _lib.RekfOptions()
        o.odom_model = int(options.odom_model)
        o.use_imu = 1 if options.use_imu else 0
        o.init_time = float(options.init_time)
        for k in range(3):
            o.init_pose[k] = float(options.init_pose[k])
        o.linear_velocity_cov = float(options.linear_velocity_cov)
        o.angular_velocity_cov = float(options.angular_velocity_cov)
        o.observation_cov = float(options.observation_cov)
        h = C.c_void_p()
        rc = self._L.rekf_create(C.byref(o), int(max_landmarks), int(device), C.byref(h))
        if rc != 0:
            raise RekfError(rc, "rekf_create")
        self._h = h
        self._cap0 = int(max_landmarks)
        # max_landmarks is the INITIAL capacity: like the reference (which resizes on every augment and never drops a reflector,
        # cc:316-363) the filter grows on demand -- the default of every wrapper (C++ EkfSlam, node_replay, this class); pass
        # auto_grow=False for a fixed capacity (overflow = sticky REKF_FLAGBIT_CAPACITY, the extra reflectors dropped)
        self.set_auto_grow(bool(auto_grow))
        self._map = load_map_txt(options.map_path)          # cc:36
        if self._map.reflector_map_.shape[0] > 0:
            self.SetGlobalMap(self._map)

    # -- lifetime -----------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.rekf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, where):
        if rc != 0:
            raise RekfError(rc, where, self._L.rekf_last_hip_error(self._h).decode())

    # -- capacity (the reference grows mu / sigma on every augment, cc:316-363) ----
    @property
    def max_landmarks(self) -> int:
        c = C.c_int()
        self._chk(self._L.rekf_get_capacity(self._h, C.byref(c)), "get_capacity")
        return c.value

    def reserve(self, max_landmarks: int):
        """Re-lay the device state out for at least ``max_landmarks`` reflectors (no-op if there is room already)."""
        self._chk(self._L.rekf_reserve(self._h, int(max_landmarks)), "reserve")

    def set_exclusive(self, on: bool = True):
        """The caller promises that this handle has the GPU to itself (include/rekf.h, rekf_set_exclusive): a scan's front end then runs
        inside the scan's own launch.  Off by default; same results either way."""
        self._chk(self._L.rekf_set_exclusive(self._h, 1 if on else 0), "set_exclusive")

    def set_auto_grow(self, on: bool = True):
        """Double the capacity whenever a scan could overflow it, instead of dropping reflectors (sticky capacity flag)."""
        self._chk(self._L.rekf_set_auto_grow(self._h, 1 if on else 0), "set_auto_grow")

    # -- reference interface --------------------------------------------------
    def HandleOdometryMessage(self, odometry: OdometryData):
        self._chk(self._L.rekf_handle_odometry(self._h, float(odometry.time), float(odometry.linear_velocity[0]),
                                               float(odometry.linear_velocity[1]),
                                               float(odometry.angular_velocity[2])), "HandleOdometryMessage")

    def HandleImuMessage(self, imu):
        return None   # empty in the reference too (reflector_ekf_slam.cc:224-227)

    def HandleObservationMessage(self, observation: Observation):
        self.handle_observation(observation.time_, observation.cloud_, observation.gps_pose_)

    def PredictState(self, time: float) -> State:
        """PredictState as the interface returns it (ekf_slam_interface.h:59, cc:97-152): the full predicted
        State -- all n means, the n x n covariance with rows/columns 0, 1 and the pose block propagated -- and,
        like the reference's `State result = state_`, the state's own (unchanged) time.  Non-mutating.  Costs an
        n x n device-to-host copy like GetState; `PredictPose` is the 96-byte fast path."""
        n = self.n
        mu = np.zeros(n)
        sig = np.zeros((n, n), order="F")
        t = C.c_double()
        nn = C.c_int()
        self._chk(self._L.rekf_predict_state_full(self._h, float(time), C.byref(t), C.byref(nn),
                                                  mu.ctypes.data_as(C.c_void_p), n, sig.ctypes.data_as(C.c_void_p), n * n),
                  "PredictState")
        return State(t.value, mu, sig)

    def PredictPose(self, time: float) -> State:
        """Pose block of PredictState only (what src/ros_node.cc:455-470 reads): mu (3,), sigma (3, 3)."""
        mu3 = (C.c_double * 3)()
        s9 = (C.c_double * 9)()
        self._chk(self._L.rekf_predict_state(self._h, float(time), mu3, s9), "PredictState")
        return State(float(time), np.array(mu3[:]), np.array(s9[:]).reshape(3, 3).T.copy())

    def flags(self) -> int:
        """Sticky device-side condition bits (REKF_FLAGBIT_CAPACITY = 1, REKF_FLAGBIT_SINGULAR = 2, REKF_FLAGBIT_STARVED = 4), not cleared."""
        f = C.c_int()
        self._chk(self._L.rekf_get_flags(self._h, C.byref(f)), "get_flags")
        return f.value

    def GetStateVector(self) -> np.ndarray:
        return self.GetState().mu

    def GetCoviarance(self) -> np.ndarray:
        return self.GetState().sigma

    def GetLatestTime(self) -> float:
        t = C.c_double()
        self._chk(self._L.rekf_get_time(self._h, C.byref(t)), "GetLatestTime")
        return t.value

    def GetState(self) -> State:
        n = self.n
        mu = np.zeros(n)
        sig = np.zeros((n, n), order="F")
        t = C.c_double()
        nn = C.c_int()
        self._chk(self._L.rekf_get_state(self._h, C.byref(t), C.byref(nn), mu.ctypes.data_as(C.c_void_p), n,
                                         sig.ctypes.data_as(C.c_void_p), n * n), "GetState")
        return State(t.value, mu, sig)

    def GetGlobalMap(self) -> Map:
        return self._map

    # -- snake_case fast paths (no Eigen-shaped copies) ------------------------
    def handle_odometry(self, t, vx, vy, wz):
        self._chk(self._L.rekf_handle_odometry(self._h, float(t), float(vx), float(vy), float(wz)),
                  "handle_odometry")

    def handle_observation(self, t, cloud, gps_pose=None):
        # (the per-scan path: a float32 C-contiguous array goes through as it is, by address -- marshalling was 3 us of a 36 us scan-to-pose)
        if not (type(cloud) is np.ndarray and cloud.dtype == np.float32 and cloud.flags.c_contiguous):
            cloud = np.ascontiguousarray(cloud, dtype=np.float32)
        if cloud.size & 1:
            raise ValueError("observations are (x, y) pairs")
        gp = None
        if gps_pose is not None:
            g = np.ascontiguousarray(gps_pose, dtype=np.float64)
            gp = g.ctypes.data
        rc = self._L.rekf_handle_observation(self._h, t, cloud.ctypes.data, cloud.size >> 1, gp)
        if rc != 0:
            self._chk(rc, "HandleObservationMessage")

    def SetGlobalMap(self, m: Map):
        xy = np.ascontiguousarray(m.reflector_map_, dtype=np.float32).reshape(-1, 2)
        cov = np.ascontiguousarray(m.reflector_map_coviarance_, dtype=np.float64).reshape(-1, 4)
        self._chk(self._L.rekf_set_map(self._h, xy.ctypes.data_as(C.c_void_p), cov.ctypes.data_as(C.c_void_p),
                                       xy.shape[0]), "SetGlobalMap")
        self._map = Map(xy.copy(), cov.reshape(-1, 2, 2).copy())

    def set_map(self, xy, cov):
        self.SetGlobalMap(Map(np.asarray(xy, np.float32).reshape(-1, 2), np.asarray(cov, np.float64).reshape(-1, 2, 2)))

    @property
    def n(self) -> int:
        n = C.c_int()
        self._chk(self._L.rekf_get_n(self._h, C.byref(n)), "get_n")
        return n.value

    def pose(self):
        b = self.__dict__.get("_pose_buf")
        if b is None:                                   # time, mu[3], sigma[9] (column-major), reused by every call
            b = self._pose_buf = np.zeros(13)
            a = b.ctypes.data
            self._pose_ptr = (a, a + 8, a + 32)
        p = self._pose_ptr
        rc = self._L.rekf_get_pose(self._h, p[0], p[1], p[2])
        if rc != 0:
            self._chk(rc, "get_pose")
        return float(b[0]), b[1:4].copy(), b[4:13].reshape(3, 3).T.copy()

    def marker_ellipses(self, max_landmarks: int | None = None) -> np.ndarray:
        """Node::ReflectorToRosMarkers' per-landmark numbers (src/ros_node.cc:750-765), computed on the device:
        (L, 5) = mx, my, angle, x_len, y_len.  40 KB D2H at 1024 landmarks instead of the n x n GetState()."""
        cap = self.max_landmarks if max_landmarks is None else int(max_landmarks)
        out = np.zeros((max(cap, 1), 5))
        k = C.c_int()
        self._chk(self._L.rekf_get_marker_ellipses(self._h, out.ctypes.data_as(C.c_void_p), cap, C.byref(k)),
                  "get_marker_ellipses")
        return out[:k.value].copy()

    def mu(self) -> np.ndarray:
        n = self.n
        mu = np.zeros(n)
        self._chk(self._L.rekf_get_state(self._h, None, None, mu.ctypes.data_as(C.c_void_p), n, None, 0), "get_mu")
        return mu

    def set_state(self, t, mu, sigma, vt=None):
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        n = mu.shape[0]
        flat = np.ascontiguousarray(np.asarray(sigma, dtype=np.float64).T).reshape(-1)   # column-major bytes
        v = None
        if vt is not None:
            vv = np.ascontiguousarray(vt, dtype=np.float64)
            v = vv.ctypes.data_as(C.c_void_p)
        self._chk(self._L.rekf_set_state(self._h, float(t), n, mu.ctypes.data_as(C.c_void_p),
                                         flat.ctypes.data_as(C.c_void_p), v), "set_state")

    def last_match(self) -> ReflectorMatchResult:
        sp = np.zeros((_lib.MAX_OBS, 2), np.int32)
        mp = np.zeros((_lib.MAX_OBS, 2), np.int32)
        nw = np.zeros((_lib.MAX_OBS,), np.int32)
        ns, nm, nn = C.c_int(), C.c_int(), C.c_int()
        self._chk(self._L.rekf_get_last_match(self._h, C.byref(ns), sp.ctypes.data_as(C.c_void_p), C.byref(nm),
                                              mp.ctypes.data_as(C.c_void_p), C.byref(nn),
                                              nw.ctypes.data_as(C.c_void_p)), "get_last_match")
        return ReflectorMatchResult(mp[: nm.value].copy(), sp[: ns.value].copy(), nw[: nn.value].copy())

    def sync(self):
        self._chk(self._L.rekf_sync(self._h), "sync")

    def sync_code(self) -> int:
        return self._L.rekf_sync(self._h)

    # -- measurement hooks -----------------------------------------------------
    def profile(self, on, only=None):
        """Bracket kernel launches with hipEvents.  ``only`` = iterable of kernel names to restrict to."""
        mask = 0
        if on:
            mask = -1 if only is None else sum(1 << KERNELS[k] for k in only)
        self._chk(self._L.rekf_profile_enable(self._h, mask), "profile_enable")

    def profile_reset(self):
        self._chk(self._L.rekf_profile_reset(self._h), "profile_reset")

    def profile_read(self) -> dict:
        out = {}
        for name, k in KERNELS.items():
            us, cnt = C.c_double(), C.c_long()
            self._chk(self._L.rekf_profile_read(self._h, k, C.byref(us), C.byref(cnt)), "profile_read")
            out[name] = (us.value, cnt.value)
        return out

    def profile_update_samples(self) -> np.ndarray:
        """Every REKF_K_UPDATE reading (us) since the last profile_reset: one hipEvent pair around each whole
        HandleObservationMessage chain on the handle's stream."""
        cnt = C.c_long()
        self._chk(self._L.rekf_profile_samples(self._h, None, 0, C.byref(cnt)), "profile_samples")
        out = np.zeros(max(cnt.value, 1), np.float32)
        self._chk(self._L.rekf_profile_samples(self._h, out.ctypes.data_as(C.c_void_p), cnt.value, C.byref(cnt)),
                  "profile_samples")
        return out[: cnt.value].astype(np.float64)

    def inject_failure(self, stage: int):
        """Test hook (include/rekf_debug.h): the next HandleObservationMessage fails at `stage` as if a HIP call had."""
        self._chk(self._L.rekf_debug_inject_failure(self._h, int(stage)), "rekf_debug_inject_failure")

    def debug_set_grid(self, on: bool = True, drift_limit: float = 0.0, mask: int = -1):
        """Test hook (include/rekf_debug.h): the match grid off / with another drift bound / with a smaller hash table."""
        self._chk(self._L.rekf_debug_set_grid(self._h, 1 if on else 0, float(drift_limit), int(mask)), "rekf_debug_set_grid")

    def time_kernel(self, name: str, reps: int = 200, ablate: int = 0) -> float:
        """Average device time (us) of `reps` back-to-back launches of one kernel of the chain between ONE
        pair of hipEvents on the handle's stream (rekf_debug_time_kernel).  Leaves the state meaningless."""
        us = C.c_double()
        self._chk(self._L.rekf_debug_time_kernel(self._h, KERNELS[name], int(reps), int(ablate), C.byref(us)),
                  "rekf_debug_time_kernel")
        return us.value

    def device_layout(self):
        ld, nmax = C.c_int(), C.c_int()
        p, m = C.c_void_p(), C.c_void_p()
        self._chk(self._L.rekf_device_layout(self._h, C.byref(ld), C.byref(nmax), C.byref(p), C.byref(m)), "layout")
        return ld.value, nmax.value, p.value, m.value
