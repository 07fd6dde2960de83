"""Deterministic synthetic worlds and sessions for the EKF hot path.

The reference ships no replayable data (its demo bag is a missing blob and ROS is
absent), so BASELINE.json's configs 2-5 are defined on synthetic sessions.  This
module generates them: a jittered grid of reflectors, a lawn-mower trajectory that
brings every reflector into range, wheel odometry at ``odom_hz`` with the launch
file's velocity noise (/root/reference/launch/slam.launch:21-22) and one
observation set per scan (the ``obs_per_scan`` nearest reflectors inside the
detector's range gate, /root/reference/src/ros_node.cc:258-265) in the robot
frame, float32 like ``sensor::PointCloud`` (/root/reference/include/sensor/sensor_data.h:15).

The same byte stream feeds the HIP path and the CPU oracle.  Everything is numpy
host code; nothing here touches the GPU.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

DIFF, OMNI = 0, 1

EV_ODOM, EV_SCAN = 0, 1


@dataclass
class SessionConfig:
    name: str
    n_landmarks: int
    obs_per_scan: int
    odom_model: int = DIFF
    seed: int = 20210330
    pitch: float = 3.0            # grid pitch (m)
    jitter: float = 0.5           # +-U(jitter) per axis
    row_spacing: float = 12.0     # lawn-mower row spacing (m)
    speed: float = 2.0            # m/s
    odom_hz: float = 50.0
    scan_hz: float = 10.0
    range_min: float = 0.3        # ros_node.cc:258-265 defaults
    range_max: float = 10.0
    sigma_v: float = 0.05         # launch/slam.launch:21  (std; the filter gets the square)
    sigma_w: float = 0.08         # launch/slam.launch:22
    sigma_obs: float = 0.05       # launch/slam.launch:23  (filter's observation std)
    obs_noise: float = 0.02       # simulated per-axis measurement noise (m)
    laps: float = 1.0             # how many times the lawn-mower route is driven
    extra_scans: int = 0          # additional scans appended after the laps (steady state)


# BASELINE.json configs[1..3] (config 0 is the missing ROS bag; config 4 = 8 x C3).
C2 = SessionConfig("C2_N128_obs16", 128, 16, DIFF, seed=20210330)
C3 = SessionConfig("C3_N1024_obs32", 1024, 32, DIFF, seed=20210331)
C4 = SessionConfig("C4_N512_omni", 512, 32, OMNI, seed=20210332)
# C4's observations come from the 3D detector run on synthetic clouds (make_point_cloud(..., **C4_LIDAR)).  The usable
# range is the range gate of the 2D configs (range_max above: a sweep then holds about 32 detectable posts); the beam
# pattern (1 degree between rings, 0.4 degree in azimuth) keeps the hits of one post within the detector's 0.2 m
# cluster tolerance of each other out to that range -- so that ONE post is ONE cluster, not one per ring -- and within
# its 160-point cluster size limit down to about 2.5 m (point_cloud_reflector_detect.cc:65-74).
C4_LIDAR = dict(rings=16, n_az=900, elev_deg=7.5, max_range=10.0)


@dataclass
class Session:
    config: SessionConfig
    landmarks: np.ndarray            # (L,2) float64 ground truth
    init_pose: np.ndarray            # (3,)
    init_time: float
    ev_type: np.ndarray              # (E,) uint8  EV_ODOM / EV_SCAN
    ev_time: np.ndarray              # (E,) float64
    odom: np.ndarray                 # (E,3) float64 (vx, vy, wz); zeros for scans
    obs_off: np.ndarray              # (E+1,) int64 offsets into obs (equal for odom events)
    obs: np.ndarray                  # (sum K,2) float32 robot-frame observations
    obs_truth_id: np.ndarray         # (sum K,) int32 ground-truth reflector id
    true_pose: np.ndarray            # (E,3) float64 true pose after each event
    meta: dict = field(default_factory=dict)

    @property
    def n_events(self) -> int:
        return int(self.ev_type.shape[0])

    def scan_indices(self) -> np.ndarray:
        return np.nonzero(self.ev_type == EV_SCAN)[0]

    def obs_of(self, e: int) -> np.ndarray:
        return self.obs[self.obs_off[e]: self.obs_off[e + 1]]


def make_world(cfg: SessionConfig, rng: np.random.Generator) -> np.ndarray:
    side = int(math.ceil(math.sqrt(cfg.n_landmarks)))
    ij = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 2)
    ij = ij[: cfg.n_landmarks].astype(np.float64)
    xy = (ij + 0.5) * cfg.pitch + rng.uniform(-cfg.jitter, cfg.jitter, size=ij.shape)
    return xy


def _route(cfg: SessionConfig, landmarks: np.ndarray) -> np.ndarray:
    lo = landmarks.min(0) - 1.0
    hi = landmarks.max(0) + 1.0
    ys = np.arange(lo[1] + cfg.row_spacing / 2, hi[1] + cfg.row_spacing / 2, cfg.row_spacing)
    ys = ys[ys < hi[1] + cfg.row_spacing / 2 - 1e-9]
    if ys.size == 0:
        ys = np.array([(lo[1] + hi[1]) / 2])
    ys = np.minimum(ys, hi[1])
    pts = []
    for r, y in enumerate(ys):
        xs = (lo[0] + 2.0, hi[0] - 2.0) if r % 2 == 0 else (hi[0] - 2.0, lo[0] + 2.0)
        pts.append((xs[0], y))
        pts.append((xs[1], y))
    # return leg closes the loop, then the robot parks in the middle of the field
    # (so that steady-state scans see a full set of obs_per_scan reflectors)
    pts.append((pts[-1][0], ys[0]))
    if pts[-1] != pts[0]:
        pts.append(pts[0])
    pts.append((float((lo[0] + hi[0]) / 2), float((lo[1] + hi[1]) / 2)))
    return np.array(pts, dtype=np.float64)


def make_session(cfg: SessionConfig, max_scans: int | None = None) -> Session:
    """Simulate the robot and return the ordered odometry/scan event stream.

    ``max_scans`` truncates the session (used by small parity tests)."""
    rng = np.random.Generator(np.random.PCG64(cfg.seed))
    lms = make_world(cfg, rng)
    route = _route(cfg, lms)
    pose = np.array([route[0, 0], route[0, 1], 0.0])
    init_pose = pose.copy()
    dt = 1.0 / cfg.odom_hz
    odom_per_scan = int(round(cfg.odom_hz / cfg.scan_hz))
    route_len = float(np.sum(np.linalg.norm(np.diff(route, axis=0), axis=1)))
    # 1.1x: the controller slows down in turns; once parked the robot stands still
    n_scans_route = int(1.1 * cfg.laps * route_len / cfg.speed * cfg.scan_hz)
    n_scans = n_scans_route + cfg.extra_scans
    if max_scans is not None:
        n_scans = min(n_scans, max_scans)

    ev_type, ev_time, odom, true_pose = [], [], [], []
    obs_chunks, id_chunks, obs_off = [], [], [0]
    t = 0.0
    wp = 1
    w_max = 1.0
    k_heading = 2.0
    for s in range(n_scans):
        for _ in range(odom_per_scan):
            # pure-pursuit style controller on the waypoint route
            tgt = route[min(wp, len(route) - 1)]
            d = tgt - pose[:2]
            if np.hypot(*d) < 1.0 and wp < len(route) - 1:
                wp += 1
                tgt = route[wp]
                d = tgt - pose[:2]
            parked = wp == len(route) - 1 and np.hypot(*d) < 0.5
            err = math.atan2(d[1], d[0]) - pose[2]
            err = math.atan2(math.sin(err), math.cos(err))
            w = max(-w_max, min(w_max, k_heading * err))
            v = cfg.speed * max(0.2, math.cos(err))
            vy = 0.0
            if cfg.odom_model == OMNI:
                vy = 0.3 * math.sin(0.2 * t)
            if parked:
                v, vy, w = 0.0, 0.0, 0.0
            # true motion: the same integrator the filter linearises
            if cfg.odom_model == DIFF:
                half = pose[2] + w * dt / 2
                pose = pose + np.array([v * dt * math.cos(half), v * dt * math.sin(half), w * dt])
            else:
                th = pose[2]
                pose = pose + np.array([v * dt * math.cos(th) - vy * dt * math.sin(th),
                                        v * dt * math.sin(th) + vy * dt * math.cos(th), w * dt])
            pose[2] = math.atan2(math.sin(pose[2]), math.cos(pose[2]))
            t += dt
            ev_type.append(EV_ODOM)
            ev_time.append(t)
            odom.append((v + rng.normal(0, cfg.sigma_v),
                         (vy + rng.normal(0, cfg.sigma_v)) if cfg.odom_model == OMNI else 0.0,
                         w + rng.normal(0, cfg.sigma_w)))
            true_pose.append(pose.copy())
            obs_off.append(obs_off[-1])
        # scan at the current time: the K nearest reflectors inside the range gate
        rel = lms - pose[:2]
        dist = np.hypot(rel[:, 0], rel[:, 1])
        cand = np.nonzero((dist >= cfg.range_min) & (dist <= cfg.range_max))[0]
        cand = cand[np.argsort(dist[cand], kind="stable")][: cfg.obs_per_scan]
        c, sn = math.cos(pose[2]), math.sin(pose[2])
        rx = c * rel[cand, 0] + sn * rel[cand, 1]
        ry = -sn * rel[cand, 0] + c * rel[cand, 1]
        order = np.argsort(np.arctan2(ry, rx), kind="stable")   # beam order, like a scan
        cand, rx, ry = cand[order], rx[order], ry[order]
        meas = np.stack([rx, ry], -1) + rng.normal(0, cfg.obs_noise, size=(cand.size, 2))
        obs_chunks.append(meas.astype(np.float32))
        id_chunks.append(cand.astype(np.int32))
        ev_type.append(EV_SCAN)
        ev_time.append(t)
        odom.append((0.0, 0.0, 0.0))
        true_pose.append(pose.copy())
        obs_off.append(obs_off[-1] + cand.size)

    obs = np.concatenate(obs_chunks) if obs_chunks else np.zeros((0, 2), np.float32)
    ids = np.concatenate(id_chunks) if id_chunks else np.zeros((0,), np.int32)
    return Session(
        config=cfg, landmarks=lms, init_pose=init_pose, init_time=0.0,
        ev_type=np.array(ev_type, dtype=np.uint8), ev_time=np.array(ev_time, dtype=np.float64),
        odom=np.array(odom, dtype=np.float64).reshape(-1, 3),
        obs_off=np.array(obs_off, dtype=np.int64), obs=obs, obs_truth_id=ids,
        true_pose=np.array(true_pose, dtype=np.float64).reshape(-1, 3),
        meta={"route_len_m": route_len, "n_scans_route": n_scans_route},
    )


def steady_state_scans(session: Session, n: int, seed_offset: int = 1000) -> list[tuple[float, np.ndarray]]:
    """``n`` extra (time, observations) scans taken from the session's final true
    pose (robot standing still), for timing steady-state updates: every observed
    reflector is already in the state, so no augment happens."""
    cfg = session.config
    rng = np.random.Generator(np.random.PCG64(cfg.seed + seed_offset))
    pose = session.true_pose[-1]
    lms = session.landmarks
    rel = lms - pose[:2]
    dist = np.hypot(rel[:, 0], rel[:, 1])
    cand = np.nonzero((dist >= cfg.range_min) & (dist <= cfg.range_max))[0]
    cand = cand[np.argsort(dist[cand], kind="stable")][: cfg.obs_per_scan]
    c, sn = math.cos(pose[2]), math.sin(pose[2])
    rx = c * rel[cand, 0] + sn * rel[cand, 1]
    ry = -sn * rel[cand, 0] + c * rel[cand, 1]
    order = np.argsort(np.arctan2(ry, rx), kind="stable")
    rx, ry = rx[order], ry[order]
    t0 = float(session.ev_time[-1]) if session.n_events else 0.0
    out = []
    for k in range(n):
        meas = np.stack([rx, ry], -1) + rng.normal(0, cfg.obs_noise, size=(rx.size, 2))
        out.append((t0 + (k + 1) / cfg.scan_hz, meas.astype(np.float32)))
    return out


# ---------------------------------------------------------------------------------------------
# raw sensor synthesis for the detector paths ("2D-detect variant")
# ---------------------------------------------------------------------------------------------
def make_laser_scan(landmarks: np.ndarray, pose, stamp: float, rng: np.random.Generator, n_beams: int = 3600,
                    sensor_xy=(0.13686, 0.0), reflector_width: float = 0.18, scan_time: float = 0.1,
                    room_half: float = 60.0, range_noise: float = 0.005, max_range: float = 30.0):
    """One 360-degree LaserScan taken at ``pose`` (x, y, theta of base_link): reflectors are flat
    plates of ``reflector_width`` facing the sensor (intensity 200), everything else is a distant
    wall (intensity 50).  Returns a dict with the sensor_msgs::LaserScan fields (float32 arrays).
    Sensor mounted at ``sensor_xy`` in base_link (launch/slam.launch:27), yaw 0."""
    x, y, th = pose
    c, s = math.cos(th), math.sin(th)
    sx = x + c * sensor_xy[0] - s * sensor_xy[1]
    sy = y + s * sensor_xy[0] + c * sensor_xy[1]
    inc = np.float32(2.0 * math.pi / n_beams)
    angle_min = np.float32(-math.pi)
    ang = angle_min + inc * np.arange(n_beams, dtype=np.float64)      # close enough for synthesis
    ranges = np.full(n_beams, np.float32(max_range * 0.8), dtype=np.float64)
    ranges += rng.normal(0, 0.05, size=n_beams)
    inten = np.full(n_beams, 50.0) + rng.normal(0, 5.0, size=n_beams)
    rel = landmarks - np.array([sx, sy])
    dist = np.hypot(rel[:, 0], rel[:, 1])
    bearing = np.arctan2(rel[:, 1], rel[:, 0]) - th
    bearing = np.arctan2(np.sin(bearing), np.cos(bearing))
    order = np.argsort(-dist)                        # nearer reflectors overwrite farther ones
    for j in order:
        if dist[j] > max_range * 0.7 or dist[j] < 0.2:
            continue
        half = math.atan2(reflector_width / 2.0, dist[j])
        lo = int(math.ceil((bearing[j] - half - float(angle_min)) / float(inc)))
        hi = int(math.floor((bearing[j] + half - float(angle_min)) / float(inc)))
        for b in range(lo, hi + 1):
            bb = b % n_beams
            delta = ang[bb] - bearing[j]
            delta = math.atan2(math.sin(delta), math.cos(delta))
            r = dist[j] / max(math.cos(delta), 1e-3)
            if r < ranges[bb]:
                ranges[bb] = r + rng.normal(0, range_noise)
                inten[bb] = 200.0 + rng.normal(0, 5.0)
    return dict(stamp=float(stamp), angle_min=float(angle_min), angle_max=float(angle_min + inc * (n_beams - 1)),
                angle_increment=float(inc), scan_time=float(scan_time), range_min=0.05, range_max=float(max_range),
                ranges=ranges.astype(np.float32), intensities=inten.astype(np.float32))


def make_point_cloud(landmarks: np.ndarray, pose, rng: np.random.Generator, rings: int = 16, n_az: int = 1800,
                     sensor_height: float = 0.7, post_radius: float = 0.09, post_z=(0.2, 1.2),
                     max_range: float = 25.0, n_outliers: int = 40, elev_deg: float = 15.0):
    """One XYZI sweep of a `rings` x `n_az` spinning lidar at base_link `pose` (sensor at the base_link
    origin): reflector posts (vertical strips, intensity 200+-20) and a dim
    background (ground / far wall, intensity 20+-10), plus `n_outliers` isolated bright points that
    the statistical outlier removal must reject.  Returns float32 (N, 4) in the sensor frame."""
    x, y, th = pose
    rel = landmarks - np.array([x, y])
    c, s = math.cos(th), math.sin(th)
    lx = c * rel[:, 0] + s * rel[:, 1]
    ly = -s * rel[:, 0] + c * rel[:, 1]
    dist = np.hypot(lx, ly)
    bearing = np.arctan2(ly, lx)
    az = -math.pi + 2.0 * math.pi * np.arange(n_az) / n_az
    elev = np.deg2rad(np.linspace(-elev_deg, elev_deg, rings))
    near = np.argsort(dist)
    near = near[(dist[near] < max_range) & (dist[near] > 0.5)]
    hit_range = np.full(n_az, np.inf)
    for j in near[::-1]:                                   # nearer posts overwrite farther ones
        half = math.asin(min(post_radius / dist[j], 1.0))
        lo = int(math.ceil((bearing[j] - half + math.pi) / (2 * math.pi) * n_az))
        hi = int(math.floor((bearing[j] + half + math.pi) / (2 * math.pi) * n_az))
        for b in range(lo, hi + 1):
            hit_range[b % n_az] = dist[j]
    pts = []
    for e in elev:
        rho = np.where(np.isfinite(hit_range), hit_range, np.nan)
        zhit = sensor_height + rho * math.tan(e)
        post = np.isfinite(hit_range) & (zhit >= post_z[0]) & (zhit <= post_z[1])
        if e < -1e-3:
            ground = sensor_height / math.tan(-e)
        else:
            ground = 40.0
        rr = np.where(post, hit_range, min(ground, 40.0))
        rr = rr + rng.normal(0, 0.01, size=n_az)
        inten = np.where(post, 200.0 + rng.normal(0, 20.0, size=n_az), 20.0 + rng.normal(0, 10.0, size=n_az))
        px = rr * np.cos(az) * 1.0
        py = rr * np.sin(az) * 1.0
        pz = np.where(post, rr * math.tan(e), np.where(e < -1e-3, -sensor_height, rr * math.tan(e)))
        pts.append(np.stack([px, py, pz, inten], -1))
    cloud = np.concatenate(pts)
    if n_outliers > 0:
        o = np.stack([rng.uniform(-15, 15, n_outliers), rng.uniform(-15, 15, n_outliers),
                      rng.uniform(-0.5, 2.0, n_outliers), np.full(n_outliers, 220.0)], -1)
        idx = rng.choice(cloud.shape[0], n_outliers, replace=False)
        cloud[idx] = o
    return cloud.astype(np.float32)
