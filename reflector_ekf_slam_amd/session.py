"""ROS-free session replay: drives a filter with the event stream of a
``synth.Session`` the way the reference's Node callbacks do.

* odometry event  -> HandleOdometryMessage            (src/ros_node.cc:627-660)
* scan event      -> HandleObservationMessage         (src/ros_node.cc:510-558)
* the FIRST scan only constructs the EKF and is dropped (src/ros_node.cc:424-441, Q11);
  here the filter is constructed by the caller with ``init_time`` = session.init_time,
  and ``replay`` skips the first scan's observations.

Works with anything exposing ``handle_odometry(t, vx, vy, wz)`` and
``handle_observation(t, cloud)`` (the HIP path's ReflectorEKFSLAM, or -- in tests --
the CPU oracle).
"""
from __future__ import annotations

from . import synth


def options_for(session: synth.Session):
    from .ekf_slam import EKFOptions
    cfg = session.config
    return EKFOptions(use_imu=False, init_time=session.init_time, init_pose=tuple(session.init_pose),
                      odom_model=cfg.odom_model, linear_velocity_cov=cfg.sigma_v ** 2,
                      angular_velocity_cov=cfg.sigma_w ** 2, observation_cov=cfg.sigma_obs ** 2)


def replay(session: synth.Session, ekf, start: int = 0, stop: int | None = None, on_scan=None,
           drop_first_scan: bool = True) -> int:
    """Feed events [start, stop) to ``ekf``.  Returns the number of scans processed.
    ``on_scan(event_index, scan_number)`` is called after each processed scan."""
    stop = session.n_events if stop is None else stop
    first = drop_first_scan and start == 0
    scans = 0
    ev_type, ev_time, odom = session.ev_type, session.ev_time, session.odom
    for e in range(start, stop):
        if ev_type[e] == synth.EV_ODOM:
            ekf.handle_odometry(ev_time[e], odom[e, 0], odom[e, 1], odom[e, 2])
        else:
            if first:
                first = False
                continue
            ekf.handle_observation(ev_time[e], session.obs_of(e))
            scans += 1
            if on_scan is not None:
                on_scan(e, scans)
    return scans
