"""Shared test plumbing: replay a golden case / synthetic session through any filter
with the snake_case interface (the CPU oracle or the HIP path)."""
from __future__ import annotations

import numpy as np

from reflector_ekf_slam_amd import synth


def norm_match(m):
    """-> (state_pairs, map_pairs, new_ids) int arrays, whatever the filter returns."""
    if isinstance(m, tuple):
        sp, mp, nw = m
    else:
        sp, mp, nw = m.state_obs_match_ids, m.map_obs_match_ids, m.new_ids
    return (np.asarray(sp, np.int32).reshape(-1, 2), np.asarray(mp, np.int32).reshape(-1, 2),
            np.asarray(nw, np.int32).reshape(-1))


def make_oracle(odom_model, init_time, init_pose, lin, ang, obs, literal=False):
    from oracle.binding import OracleEKF
    return OracleEKF(odom_model, init_time, init_pose, lin, ang, obs, literal=literal)


def make_gpu(odom_model, init_time, init_pose, lin, ang, obs, max_landmarks):
    from reflector_ekf_slam_amd import EKFOptions, ReflectorEKFSLAM
    opt = EKFOptions(use_imu=False, init_time=float(init_time), init_pose=tuple(float(v) for v in init_pose),
                     odom_model=int(odom_model), linear_velocity_cov=float(lin), angular_velocity_cov=float(ang),
                     observation_cov=float(obs))
    return ReflectorEKFSLAM(opt, max_landmarks=max_landmarks, device=0, auto_grow=False)


def replay_golden(g, filt, check_every=1):
    """Replays golden case `g` (an np.load result) through `filt` and compares per scan
    with the expected records.  Returns (max pose err, max |n| mismatch count, match mismatches)."""
    first = True
    k = 0
    worst_pose = 0.0
    bad_match = 0
    bad_n = 0
    worst_tr = 0.0
    has_map = g["map_xy"].shape[0] > 0
    if has_map:
        filt.set_map(g["map_xy"], g["map_cov"]) if hasattr(filt, "set_map") else None
    for e in range(g["ev_type"].shape[0]):
        if g["ev_type"][e] == synth.EV_ODOM:
            filt.handle_odometry(g["ev_time"][e], *g["odom"][e])
            continue
        if first:
            first = False
            continue
        ob = g["obs"][g["obs_off"][e]: g["obs_off"][e + 1]]
        gps = g["gps"][e]
        filt.handle_observation(g["ev_time"][e], ob, None if np.isnan(gps[0]) else gps)
        if k % check_every == 0:
            sp, mp, nw = norm_match(filt.last_match())
            es = g["exp_state"][g["exp_state_off"][k]: g["exp_state_off"][k + 1]].reshape(-1, 2)
            em = g["exp_map"][g["exp_map_off"][k]: g["exp_map_off"][k + 1]].reshape(-1, 2)
            en = g["exp_new"][g["exp_new_off"][k]: g["exp_new_off"][k + 1]]
            if not (np.array_equal(sp, es) and np.array_equal(mp, em) and np.array_equal(nw, en)):
                bad_match += 1
            mu = filt.mu()
            if mu.shape[0] != g["exp_n"][k]:
                bad_n += 1
            worst_pose = max(worst_pose, float(np.abs(mu[:3] - g["exp_pose"][k]).max()))
        k += 1
    return worst_pose, bad_n, bad_match


def drive_pair(sess, a, b, on_scan=None, max_events=None):
    """Feeds the same session to filters a and b in lock-step."""
    first = True
    scans = 0
    E = sess.n_events if max_events is None else min(max_events, sess.n_events)
    for e in range(E):
        if sess.ev_type[e] == synth.EV_ODOM:
            a.handle_odometry(sess.ev_time[e], *sess.odom[e])
            b.handle_odometry(sess.ev_time[e], *sess.odom[e])
        else:
            if first:
                first = False
                continue
            ob = sess.obs_of(e)
            a.handle_observation(sess.ev_time[e], ob)
            b.handle_observation(sess.ev_time[e], ob)
            scans += 1
            if on_scan is not None:
                on_scan(e, scans)
    return scans


def ellipse_matrices(ell):
    """(L,5) marker ellipses {mx,my,angle,x_len,y_len} -> (L,2,2) the matrices R diag(l0,l1) R^T they draw
    (l = (len/2)^2 / 5.991): two descriptions of the same ellipse (pairs swapped, vector sign) agree here."""
    ell = np.asarray(ell, np.float64).reshape(-1, 5)
    c, s = np.cos(ell[:, 2]), np.sin(ell[:, 2])
    l0, l1 = (ell[:, 3] / 2) ** 2 / 5.991, (ell[:, 4] / 2) ** 2 / 5.991
    R = np.stack([np.stack([c, -s], -1), np.stack([s, c], -1)], -2)
    return np.einsum("nij,nj,nkj->nik", R, np.stack([l0, l1], -1), R)
