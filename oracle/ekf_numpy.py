"""oracle/ekf_numpy.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Independent numpy restatement of the reference EKF-SLAM core
(/root/reference/src/reflector_ekf_slam/reflector_ekf_slam.cc and the pose-fusion
branch of reflector_ekf_slam_gps.cc).  It follows the Eigen expressions
*literally* (dense G, dense H, ``K = P H^T (H P H^T + Q)^-1``,
``P = P - K H P``), written separately from oracle/ekf_oracle.c so that the two
restatements pin each other (PARITY UNPINNED: the reference has no tests or
fixtures and cannot be built here -- see the header of ekf_oracle.c).

Only tests/ and tests/golden/make_golden.py import this module.
"""
from __future__ import annotations

import math

import numpy as np

DIFF, OMNI = 0, 1


class NumpyEKF:
    """Mirrors ekf::ReflectorEKFSLAM (reflector_ekf_slam.h:13-64)."""

    def __init__(self, odom_model, init_time, init_pose, lin_cov, ang_cov, obs_cov):
        # ctor: reflector_ekf_slam.cc:6-37
        self.model = DIFF if odom_model == DIFF else OMNI
        self.time = float(init_time)
        self.mu = np.array(init_pose, dtype=np.float64).copy()
        self.sigma = np.zeros((3, 3))
        self.vt = np.zeros(3)
        if self.model == DIFF:
            self.Qu = np.diag([lin_cov, ang_cov]).astype(np.float64)
        else:
            self.Qu = np.diag([lin_cov, lin_cov, ang_cov]).astype(np.float64)
        self.Qt = np.diag([obs_cov, obs_cov]).astype(np.float64)
        self.map_xy = np.zeros((0, 2), dtype=np.float32)
        self.map_cov = np.zeros((0, 2, 2))
        self.last_match = ([], [], [])

    def set_map(self, xy, cov):
        self.map_xy = np.asarray(xy, dtype=np.float32).reshape(-1, 2)
        self.map_cov = np.asarray(cov, dtype=np.float64).reshape(-1, 2, 2)

    # -- Predict / PredictState: reflector_ekf_slam.cc:154-206 / :97-152 ------
    def _motion(self, mu, sigma, dt):
        N = mu.shape[0]
        vx, vy, w = self.vt
        th = mu[2]
        G = np.eye(N)
        if self.model == DIFF:
            dth = w * dt
            half = th + dth / 2
            dx = vx * dt * math.cos(half)
            dy = vx * dt * math.sin(half)
            G[0, 2] = -vx * dt * math.sin(half)
            G[1, 2] = vx * dt * math.cos(half)
            Gu = np.zeros((N, 2))
            Gu[0:3, :] = [[dt * math.cos(half), -vx * dt * dt * math.sin(half) / 2],
                          [dt * math.sin(half), vx * dt * dt * math.cos(half) / 2],
                          [0.0, dt]]
        else:
            dth = w * dt
            dx = vx * dt * math.cos(th) - vy * dt * math.sin(th)
            dy = vx * dt * math.sin(th) + vy * dt * math.cos(th)
            G[0, 2] = -vx * dt * math.sin(th) - vy * dt * math.cos(th)
            G[1, 2] = vx * dt * math.cos(th) - vy * dt * math.sin(th)
            Gu = np.zeros((N, 3))
            Gu[0:3, :] = [[dt * math.cos(th), -dt * math.sin(th), 0.0],
                          [dt * math.sin(th), dt * math.cos(th), 0.0],
                          [0.0, 0.0, dt]]
        sigma_new = G @ sigma @ G.T + Gu @ self.Qu @ Gu.T
        mu_new = mu.copy()
        mu_new[0:3] += [dx, dy, dth]
        mu_new[2] = math.atan2(math.sin(mu_new[2]), math.cos(mu_new[2]))
        return mu_new, sigma_new

    def predict(self, dt):
        self.mu, self.sigma = self._motion(self.mu, self.sigma, dt)

    def predict_state(self, time):
        return self._motion(self.mu, self.sigma, time - self.time)

    # -- HandleOdometryMessage: reflector_ekf_slam.cc:208-223 -----------------
    def handle_odometry(self, t, vx, vy, wz):
        if t < self.time:
            return
        self.vt = np.array([vx, vy, wz], dtype=np.float64)
        self.predict(t - self.time)
        self.time = t

    # -- ReflectorMatch: reflector_ekf_slam.cc:370-455 ------------------------
    def _to_global(self, p):
        th = self.mu[2]
        x = np.float32(float(p[0]) * math.cos(th) - float(p[1]) * math.sin(th) + self.mu[0])
        y = np.float32(float(p[0]) * math.sin(th) + float(p[1]) * math.cos(th) + self.mu[1])
        return np.array([x, y], dtype=np.float32)

    def _match(self, obs):
        map_pairs, state_pairs, new_ids = [], [], []
        K = obs.shape[0]
        if self.mu.shape[0] == 3 and self.map_xy.shape[0] == 0:
            return [], [], list(range(K))
        M = (self.mu.shape[0] - 3) // 2
        M_ = self.map_xy.shape[0]
        lm32 = self.mu[3:].astype(np.float32).reshape(-1, 2)
        for i in range(K):
            g = self._to_global(obs[i])
            if M_ > 0:
                delta = (self.map_xy - g).astype(np.float32).astype(np.float64)  # float32 subtract
                d = np.sqrt(np.einsum("ji,jik,jk->j", delta, self.map_cov, delta))
                j = int(np.argmin(d))  # first minimum
                if d[j] < 0.05:
                    map_pairs.append((i, j))
                    continue
            if M > 0:
                delta = (g - lm32).astype(np.float32).astype(np.float64)
                d = np.sqrt(delta[:, 0] * delta[:, 0] + delta[:, 1] * delta[:, 1])
                j = int(np.argmin(d))
                if d[j] < 0.6:
                    state_pairs.append((i, j))
                    continue
            new_ids.append(i)
        return map_pairs, state_pairs, new_ids

    # -- HandleObservationMessage: reflector_ekf_slam.cc:229-368 --------------
    def handle_observation(self, t, obs, gps_pose=None):
        obs = np.asarray(obs, dtype=np.float32).reshape(-1, 2)
        self.predict(t - self.time)
        self.time = t
        self.last_match = ([], [], [])
        if obs.shape[0] == 0:
            return
        map_pairs, state_pairs, new_ids = self._match(obs)
        self.last_match = (map_pairs, state_pairs, new_ids)
        M, M_ = len(state_pairs), len(map_pairs)
        MM = M + M_
        N = self.mu.shape[0]
        if MM > 0:
            H = np.zeros((2 * MM, N))
            zt = np.zeros(2 * MM)
            zh = np.zeros(2 * MM)
            Q = np.zeros((2 * MM, 2 * MM))
            c, s = math.cos(self.mu[2]), math.sin(self.mu[2])
            B = np.array([[c, s], [-s, c]])
            rows = [(i, l, g, True) for i, (l, g) in enumerate(state_pairs)] + \
                   [(M + i, l, g, False) for i, (l, g) in enumerate(map_pairs)]
            for i, local_id, global_id, is_state in rows:
                zt[2 * i: 2 * i + 2] = obs[local_id].astype(np.float64)
                if is_state:
                    lx, ly = self.mu[3 + 2 * global_id], self.mu[4 + 2 * global_id]
                else:
                    lx, ly = float(self.map_xy[global_id, 0]), float(self.map_xy[global_id, 1])
                dx, dy = lx - self.mu[0], ly - self.mu[1]
                zh[2 * i] = dx * c + dy * s
                zh[2 * i + 1] = -dx * s + dy * c
                H[2 * i: 2 * i + 2, 0:3] = [[-c, -s, -dx * s + dy * c],
                                             [s, -c, -dx * c - dy * s]]
                if is_state:
                    H[2 * i: 2 * i + 2, 3 + 2 * global_id: 5 + 2 * global_id] = B
                Q[2 * i: 2 * i + 2, 2 * i: 2 * i + 2] = self.Qt
            dz = zt - zh
            if gps_pose is not None:  # reflector_ekf_slam_gps.cc:305-340
                H2 = np.zeros((2 * MM + 3, N))
                H2[: 2 * MM] = H
                H2[2 * MM:, 0:3] = np.eye(3)
                dz2 = np.zeros(2 * MM + 3)
                dz2[: 2 * MM] = dz
                dz2[2 * MM:] = np.asarray(gps_pose, dtype=np.float64) - self.mu[0:3]
                dth = dz2[2 * MM + 2]
                qw, qz = math.cos(dth / 2), math.sin(dth / 2)
                nrm = math.hypot(qw, qz)
                qw, qz = qw / nrm, qz / nrm
                if qw < 0:
                    qw, qz = -qw, -qz
                ang = 2.0 * math.atan2(abs(qz), qw)
                scale = 2.0 if ang < 1e-7 else ang / math.sin(ang / 2.0)
                dz2[2 * MM + 2] = scale * qz
                Q2 = np.zeros((2 * MM + 3, 2 * MM + 3))
                Q2[: 2 * MM, : 2 * MM] = Q
                Q2[2 * MM:, 2 * MM:] = np.diag([0.05 * 0.05, 0.05 * 0.05, 0.017 * 0.017])
                H, dz, Q = H2, dz2, Q2
            K_t = self.sigma @ H.T @ np.linalg.inv(H @ self.sigma @ H.T + Q)
            self.mu = self.mu + K_t @ dz
            self.mu[2] = math.atan2(math.sin(self.mu[2]), math.cos(self.mu[2]))
            self.sigma = self.sigma - K_t @ H @ self.sigma
        N2 = len(new_ids)
        if N2 > 0:
            Me = N + 2 * N2
            xe = np.zeros(Me)
            xe[:N] = self.mu
            Sg = np.zeros((Me, Me))
            Sg[:N, :N] = self.sigma
            Sxi = self.sigma[0:3, 0:3].copy()
            s, c = math.sin(self.mu[2]), math.cos(self.mu[2])
            Gz1 = np.array([[c, -s], [s, c]])
            Gp = np.zeros((2 * N2, 3))
            Gz = np.zeros((2 * N2, 2))
            Gfx = np.zeros((2 * N2, N))
            for i, local_id in enumerate(new_ids):
                g = self._to_global(obs[local_id])
                xe[N + 2 * i] = float(g[0])
                xe[N + 2 * i + 1] = float(g[1])
                rx, ry = float(obs[local_id, 0]), float(obs[local_id, 1])
                Gp_i = np.array([[1.0, 0.0, -rx * s - ry * c], [0.0, 1.0, rx * c - ry * s]])
                Gp[2 * i: 2 * i + 2] = Gp_i
                Gz[2 * i: 2 * i + 2] = Gz1
                Gfx[2 * i: 2 * i + 2, 0:3] = Gp_i
            Smm = Gp @ Sxi @ Gp.T + Gz @ self.Qt @ Gz.T
            Smx = Gfx @ self.sigma
            Sg[N:, :N] = Smx
            Sg[:N, N:] = Smx.T
            Sg[N:, N:] = Smm
            self.mu, self.sigma = xe, Sg
