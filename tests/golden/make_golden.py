"""Generates the committed golden vectors under tests/golden/*.npz.

There is nothing to import from the reference (it is C++ that needs ROS/Eigen/glog and
ships no fixtures), so the vectors come from the INDEPENDENT numpy restatement
oracle/ekf_numpy.py, which follows the Eigen expressions of
/root/reference/src/reflector_ekf_slam/reflector_ekf_slam.cc (+ the pose branch of
reflector_ekf_slam_gps.cc) literally.  The C oracle and the HIP path are both tested
against these files.  Run from the repo root:   python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.ekf_numpy import NumpyEKF  # noqa: E402
from reflector_ekf_slam_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (config, max_scans, with_map, with_gps)
    "diff_L24_obs8": (synth.SessionConfig("diff_L24_obs8", 24, 8, synth.DIFF, seed=7, speed=1.0, row_spacing=6.0), 160, False, False),
    "omni_L30_obs10": (synth.SessionConfig("omni_L30_obs10", 30, 10, synth.OMNI, seed=8, speed=1.0, row_spacing=6.0), 160, False, False),
    "map_L24_obs8": (synth.SessionConfig("map_L24_obs8", 24, 8, synth.DIFF, seed=9, speed=1.0, row_spacing=6.0), 120, True, False),
    "gps_L20_obs6": (synth.SessionConfig("gps_L20_obs6", 20, 6, synth.DIFF, seed=10, speed=1.0, row_spacing=6.0), 120, False, True),
    "diff_L128_obs16": (synth.C2, 260, False, False),
}


def make_case(name):
    cfg, max_scans, with_map, with_gps = CASES[name]
    s = synth.make_session(cfg, max_scans=max_scans)
    rng = np.random.Generator(np.random.PCG64(cfg.seed + 555))
    f = NumpyEKF(cfg.odom_model, s.init_time, s.init_pose, cfg.sigma_v ** 2, cfg.sigma_w ** 2, cfg.sigma_obs ** 2)
    map_xy = np.zeros((0, 2), np.float32)
    map_cov = np.zeros((0, 4))
    if with_map:
        ids = np.arange(0, cfg.n_landmarks, 3)
        map_xy = (s.landmarks[ids] + rng.normal(0, 0.01, size=(ids.size, 2))).astype(np.float32)
        # delta^T Sigma delta < 0.05^2 with Sigma = 0.01 I  <=>  |delta| < 0.5 m  (quirk Q3)
        map_cov = np.tile(np.array([0.01, 0.0, 0.0, 0.01]), (ids.size, 1))
        f.set_map(map_xy, map_cov)
    gps = np.full((s.n_events, 3), np.nan)
    poses, ns, traces, sums = [], [], [], []
    st_off, st_flat, mp_off, mp_flat, nw_off, nw_flat = [0], [], [0], [], [0], []
    first = True
    for e in range(s.n_events):
        if s.ev_type[e] == synth.EV_ODOM:
            f.handle_odometry(s.ev_time[e], *s.odom[e])
            continue
        if first:
            first = False
            continue
        g = None
        if with_gps:
            g = s.true_pose[e] + rng.normal(0, [0.03, 0.03, 0.01])
            gps[e] = g
        f.handle_observation(s.ev_time[e], s.obs_of(e), g)
        mp, sp, nw = f.last_match
        st_flat += [v for p in sp for v in p]; st_off.append(len(st_flat))
        mp_flat += [v for p in mp for v in p]; mp_off.append(len(mp_flat))
        nw_flat += list(nw); nw_off.append(len(nw_flat))
        poses.append(f.mu[:3].copy()); ns.append(f.mu.shape[0])
        traces.append(np.trace(f.sigma)); sums.append(np.abs(f.sigma).sum())
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        odom_model=cfg.odom_model, init_time=s.init_time, init_pose=s.init_pose,
        lin_cov=cfg.sigma_v ** 2, ang_cov=cfg.sigma_w ** 2, obs_cov=cfg.sigma_obs ** 2,
        n_landmarks=cfg.n_landmarks,
        ev_type=s.ev_type, ev_time=s.ev_time, odom=s.odom, obs_off=s.obs_off, obs=s.obs, gps=gps,
        map_xy=map_xy, map_cov=map_cov,
        exp_pose=np.array(poses), exp_n=np.array(ns), exp_trace=np.array(traces), exp_abs_sum=np.array(sums),
        exp_state_off=np.array(st_off), exp_state=np.array(st_flat, dtype=np.int32),
        exp_map_off=np.array(mp_off), exp_map=np.array(mp_flat, dtype=np.int32),
        exp_new_off=np.array(nw_off), exp_new=np.array(nw_flat, dtype=np.int32),
        exp_final_mu=f.mu, exp_final_sigma=f.sigma,
    )
    print(name, "scans", len(ns), "final n", ns[-1], "map matches", len(mp_flat) // 2)


if __name__ == "__main__":
    for name in (sys.argv[1:] or CASES):
        make_case(name)
