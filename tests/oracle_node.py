"""The reference's three Node components over the CPU ORACLE, shaped like the HIP path's Python mirrors, so that
reflector_ekf_slam_amd.node_replay.Node can run the same message flow over both (test infrastructure only)."""
from __future__ import annotations

import numpy as np

from reflector_ekf_slam_amd.detect import RangeData
from reflector_ekf_slam_amd.ekf_slam import Map, Observation, State, load_map_txt
from reflector_ekf_slam_amd.node_replay import Backend


class OracleDetector:
    def __init__(self, opt, s2b):
        from oracle.binding import OracleDetect2D
        self._o = OracleDetect2D(sensor_to_base_link=tuple(s2b), intensity_min=opt.intensity_min,
                                 reflector_min_length=opt.reflector_min_length, reflector_length_error=opt.reflector_length_error,
                                 range_min=opt.range_min, range_max=opt.range_max)
        self._s2b = np.asarray(s2b, np.float64)

    def HandleOdometryData(self, m):
        self._o.handle_odometry(m.time, m.position[0], m.position[1], m.orientation[3], m.orientation[0], m.linear_velocity[0],
                                m.linear_velocity[1], m.angular_velocity[2])

    def HandleLaserScan(self, scan):
        t, c = self._o.handle_scan(scan)
        return Observation(t, c)

    def GetRangeData(self):
        return RangeData(self._s2b[:2].astype(np.float32), self._o.returns())          # laser_reflector_detect.cc:243


class OracleSlam:
    def __init__(self, opt):
        from oracle.binding import OracleEKF
        self._o = OracleEKF(opt.odom_model, opt.init_time, np.asarray(opt.init_pose, np.float64), opt.linear_velocity_cov,
                            opt.angular_velocity_cov, opt.observation_cov)
        self._map = load_map_txt(opt.map_path)
        if self._map.reflector_map_.shape[0] > 0:
            self._o.set_map(self._map.reflector_map_, self._map.reflector_map_coviarance_)

    def HandleOdometryMessage(self, m):
        self._o.handle_odometry(m.time, m.linear_velocity[0], m.linear_velocity[1], m.angular_velocity[2])

    def HandleObservationMessage(self, obs):
        self._o.handle_observation(obs.time_, obs.cloud_, obs.gps_pose_)

    def pose(self):
        mu, P = self._o.state()
        return self._o.time, mu[:3].copy(), P[:3, :3].copy()

    def GetState(self):
        mu, P = self._o.state()
        return State(self._o.time, mu, P)

    def GetGlobalMap(self) -> Map:
        return self._map

    @property
    def n(self):
        return self._o.n


def oracle_backend() -> Backend:
    from reflector_ekf_slam_amd.map_builder import MapBuilder, MapBuilderOptions
    from tests.oracle_front_end import OracleFrontEnd
    return Backend(make_detector=lambda opt, s2b: OracleDetector(opt, s2b), make_ekf=lambda opt: OracleSlam(opt),
                   make_map_builder=lambda: MapBuilder(MapBuilderOptions(), front_end=OracleFrontEnd()))
