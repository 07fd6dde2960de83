"""Node-level replay of BASELINE.json configs[0] without ROS: the message flow of the reference's ``Node``
(/root/reference/src/ros_node.cc) over an on-disk dump of the two topics it subscribes to.

    odometry message  -> OdometryCallback  (src/ros_node.cc:627-660): detector.HandleOdometryData, slam.HandleOdometryMessage,
                                                                      GetState -> path
    laser scan        -> ScanCallback      (src/ros_node.cc:420-558): the FIRST scan only constructs the EKF (init_time = its
                                                                      stamp, Q11); later ones: HandleLaserScan ->
                                                                      HandleObservationMessage -> GetState ->
                                                                      MapBuilder::AddRangeData(GetRangeData(), ekf pose)
    SaveReflectorResult (src/ros_node.cc:75-140): the two-line txt map (optionally the reference's very bytes)

Dump schema ``rekf-dump-v1`` -- one numpy ``.npz`` (what a rosbag of this node holds, minus ROS):
    meta         json string: {"schema": "rekf-dump-v1", "odom_model": "diff"|"omni", "initial_pose": [x, y, yaw],
                 "sensor_to_base_link": [x, y, yaw], "linear_velocity_sigma", "angular_velocity_sigma", "observation_sigma",
                 "detector": {intensity_min, reflector_min_length, reflector_length_error, range_min, range_max},
                 "map_path": ""}                                                       (launch/slam.launch, src/ros_node.cc:150-300)
    odom_t       [No]     float64   nav_msgs/Odometry header.stamp (s)
    odom_pose    [No, 4]  float64   pose.position.x, .y, pose.orientation.z, .w
    odom_twist   [No, 3]  float64   twist.linear.x, .y, twist.angular.z
    scan_t       [Ns]     float64   sensor_msgs/LaserScan header.stamp (s)
    scan_params  [Ns, 6]  float32   angle_min, angle_max, angle_increment, scan_time, range_min, range_max
    scan_off     [Ns + 1] int64     beam offsets into ranges / intensities
    ranges, intensities  [sum of beams] float32
Messages are replayed in stamp order (ties: odometry first -- ros::spin() is single-threaded, src/ros_node.cc:68).

The three components are injected (``Backend``): the default is the HIP path (detect.LaserReflectorDetect, ReflectorEKFSLAM,
map_builder.MapBuilder); the GPU suite runs the same harness over the CPU oracle's components for comparison.
"""
from __future__ import annotations

import json
import math
from dataclasses import dataclass, field

import numpy as np

SCHEMA = "rekf-dump-v1"


# ------------------------------------------------------------------------------------------------ dump I/O
@dataclass
class Dump:
    meta: dict
    odom_t: np.ndarray
    odom_pose: np.ndarray
    odom_twist: np.ndarray
    scan_t: np.ndarray
    scan_params: np.ndarray
    scan_off: np.ndarray
    ranges: np.ndarray
    intensities: np.ndarray

    def scan(self, k: int):
        from .detect import LaserScan
        a, b = int(self.scan_off[k]), int(self.scan_off[k + 1])
        p = self.scan_params[k]
        return LaserScan(float(self.scan_t[k]), float(p[0]), float(p[1]), float(p[2]), float(p[3]), float(p[4]), float(p[5]),
                         self.ranges[a:b], self.intensities[a:b])

    def events(self):
        """(kind, index) in stamp order, odometry before a scan of the same stamp."""
        ev = [(float(t), 0, i) for i, t in enumerate(self.odom_t)] + [(float(t), 1, i) for i, t in enumerate(self.scan_t)]
        ev.sort()
        return [("odom" if k == 0 else "scan", i) for _, k, i in ev]


def write_dump(path: str, meta: dict, odom, scans) -> None:
    """odom: iterable of (t, px, py, qz, qw, vx, vy, wz); scans: iterable of LaserScan-like objects."""
    odom = np.asarray(list(odom), dtype=np.float64).reshape(-1, 8)
    scans = list(scans)
    off = np.zeros(len(scans) + 1, np.int64)
    for k, s in enumerate(scans):
        off[k + 1] = off[k] + len(s.ranges)
    m = dict(meta)
    m["schema"] = SCHEMA
    np.savez_compressed(
        path, meta=np.array(json.dumps(m)), odom_t=odom[:, 0], odom_pose=odom[:, 1:5], odom_twist=odom[:, 5:8],
        scan_t=np.array([s.stamp for s in scans], np.float64),
        scan_params=np.array([[s.angle_min, s.angle_max, s.angle_increment, s.scan_time, s.range_min, s.range_max] for s in scans],
                             np.float32).reshape(-1, 6),
        scan_off=off,
        ranges=np.concatenate([np.asarray(s.ranges, np.float32) for s in scans]) if scans else np.zeros(0, np.float32),
        intensities=np.concatenate([np.asarray(s.intensities, np.float32) for s in scans]) if scans else np.zeros(0, np.float32))


def read_dump(path: str) -> Dump:
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    if meta.get("schema") != SCHEMA:
        raise ValueError(f"{path}: not a {SCHEMA} dump (schema = {meta.get('schema')!r})")
    d = Dump(meta, z["odom_t"], z["odom_pose"], z["odom_twist"], z["scan_t"], z["scan_params"], z["scan_off"], z["ranges"],
             z["intensities"])
    if d.scan_off.shape[0] != d.scan_t.shape[0] + 1 or d.scan_off[-1] != d.ranges.shape[0] or d.ranges.shape != d.intensities.shape:
        raise ValueError(f"{path}: inconsistent scan arrays")
    return d


def synth_dump(path: str, cfg, max_scans: int = 60, n_beams: int = 1440, seed: int = 1, sensor_to_base_link=(0.13686, 0.0, 0.0)) -> Dump:
    """A synthetic bag for the harness: a ``synth`` session's odometry (true poses as the odometer's pose, its noisy twist)
    and one simulated LaserScan per scan event."""
    from types import SimpleNamespace

    from . import synth
    sess = synth.make_session(cfg, max_scans=max_scans)
    rng = np.random.Generator(np.random.PCG64(seed))
    odom, scans = [], []
    for e in range(sess.n_events):
        t = float(sess.ev_time[e])
        x, y, th = sess.true_pose[e]
        if sess.ev_type[e] == synth.EV_ODOM:
            odom.append((t, x, y, math.sin(th / 2), math.cos(th / 2), sess.odom[e][0], sess.odom[e][1], sess.odom[e][2]))
        else:
            scans.append(SimpleNamespace(**synth.make_laser_scan(sess.landmarks, sess.true_pose[e], t, rng, n_beams=n_beams,
                                                                 sensor_xy=tuple(sensor_to_base_link[:2]))))
    meta = {"odom_model": "diff" if cfg.odom_model == synth.DIFF else "omni", "initial_pose": [float(v) for v in sess.init_pose],
            "sensor_to_base_link": [float(v) for v in sensor_to_base_link],
            "linear_velocity_sigma": cfg.sigma_v, "angular_velocity_sigma": cfg.sigma_w, "observation_sigma": cfg.sigma_obs,
            "detector": {"intensity_min": 160.0, "reflector_min_length": 0.18, "reflector_length_error": 0.06, "range_min": 0.3,
                         "range_max": 10.0},
            "map_path": ""}
    write_dump(path, meta, odom, scans)
    return read_dump(path)


# ------------------------------------------------------------------------------------------------ the node
@dataclass
class Backend:
    """Factories of the three components ``Node`` owns (include/ros_node.h:131-140)."""
    make_detector: object          # (ReflectorDetectOptions, sensor_to_base_link xyyaw) -> HandleOdometryData / HandleLaserScan / GetRangeData
    make_ekf: object               # (EKFOptions) -> HandleOdometryMessage / HandleObservationMessage / pose() / GetState()
    make_map_builder: object       # () -> AddRangeData(time, RangeData, (x, y, yaw))


def hip_backend(max_landmarks: int = 64, device: int = 0, auto_grow: bool = True) -> Backend:
    from .detect import LaserReflectorDetect
    from .ekf_slam import ReflectorEKFSLAM
    from .map_builder import MapBuilder, MapBuilderOptions
    return Backend(
        make_detector=lambda opt, s2b: LaserReflectorDetect(opt, max_beams=8192, device=device, sensor_to_base_link=s2b),
        make_ekf=lambda opt: ReflectorEKFSLAM(opt, max_landmarks=max_landmarks, device=device, auto_grow=auto_grow),
        make_map_builder=lambda: MapBuilder(MapBuilderOptions(), max_points=16384, max_cells=2048 * 2048))


@dataclass
class ReplayLog:
    path: list = field(default_factory=list)                 # (stamp, x, y, theta) after every callback that publishes (ekf_path_)
    observations: list = field(default_factory=list)         # (obs time, centres) per processed scan
    match_poses: list = field(default_factory=list)          # MatchingResult.local_pose per scan (or None)


class Node:
    """The reference's ``Node`` minus ROS: same callbacks, same order of calls into the three components."""

    def __init__(self, meta: dict, backend: Backend):
        from .detect import ReflectorDetectOptions
        self.meta = meta
        self.backend = backend
        d = meta.get("detector", {})
        self.detector = backend.make_detector(ReflectorDetectOptions(**d), tuple(meta.get("sensor_to_base_link", (0.0, 0.0, 0.0))))
        self.slam = None
        self.map_builder = backend.make_map_builder()
        self.log = ReplayLog()

    def _ekf_options(self, time):
        from .ekf_slam import DIFF, OMNI, EKFOptions
        m = self.meta
        return EKFOptions(use_imu=False, init_time=float(time), init_pose=tuple(m.get("initial_pose", (0.0, 0.0, 0.0))),
                          map_path=m.get("map_path", ""), odom_model=DIFF if m.get("odom_model", "diff") == "diff" else OMNI,
                          linear_velocity_cov=float(m["linear_velocity_sigma"]) ** 2,        # the node squares the launch sigmas
                          angular_velocity_cov=float(m["angular_velocity_sigma"]) ** 2,      # (src/ros_node.cc:207-238)
                          observation_cov=float(m["observation_sigma"]) ** 2)

    def OdometryCallback(self, t, pose4, twist3):                                            # src/ros_node.cc:627-660
        from .ekf_slam import OdometryData
        px, py, qz, qw = (float(v) for v in pose4)
        vx, vy, wz = (float(v) for v in twist3)
        odom = OdometryData(float(t), (vx, vy, 0.0), (0.0, 0.0, wz), (px, py, 0.0), (qw, 0.0, 0.0, qz))
        self.detector.HandleOdometryData(odom)
        if self.slam is not None:
            self.slam.HandleOdometryMessage(odom)
            _, mu3, _ = self.slam.pose()                                                     # GetState(): the pose is all the node reads
            self.log.path.append((float(t), float(mu3[0]), float(mu3[1]), float(mu3[2])))

    def ScanCallback(self, scan):                                                            # src/ros_node.cc:420-558
        from .map_builder import RangeData
        if self.slam is None:
            self.slam = self.backend.make_ekf(self._ekf_options(scan.stamp))                 # the first scan is not processed (Q11)
            return
        obs = self.detector.HandleLaserScan(scan)
        self.slam.HandleObservationMessage(obs)
        _, mu3, _ = self.slam.pose()
        self.log.observations.append((obs.time_, np.array(obs.cloud_, np.float32)))
        if obs.cloud_.shape[0] > 0:                                                          # :524: publishes only with reflectors in view
            self.log.path.append((float(scan.stamp), float(mu3[0]), float(mu3[1]), float(mu3[2])))
        rd = self.detector.GetRangeData()
        res = self.map_builder.AddRangeData(float(scan.stamp), RangeData(rd.origin, rd.returns, np.zeros((0, 2), np.float32)),
                                            (float(mu3[0]), float(mu3[1]), float(mu3[2])))
        self.log.match_poses.append(None if res is None else np.array(res.local_pose))

    def SaveReflectorResult(self, filebase: str, reference_bytes: bool = True) -> str:       # src/ros_node.cc:75-140
        from .ekf_slam import save_map_txt
        if self.slam is None:
            return ""
        path = filebase + ".txt"
        save_map_txt(path, self.slam.GetState(), self.slam.GetGlobalMap(), reference_bytes=reference_bytes)
        return path

    def run(self, dump: Dump) -> ReplayLog:
        for kind, i in dump.events():
            if kind == "odom":
                self.OdometryCallback(dump.odom_t[i], dump.odom_pose[i], dump.odom_twist[i])
            else:
                self.ScanCallback(dump.scan(i))
        return self.log


def replay(dump: Dump, backend: Backend | None = None) -> Node:
    node = Node(dump.meta, backend or hip_backend())
    node.run(dump)
    return node
