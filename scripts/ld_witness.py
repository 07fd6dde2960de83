"""Which FP64 arithmetic is closer to exact?  Replays a session dumped by scripts/gpu_repro_seed.py (FUZZ_DUMP=file.npz: every event as fed, the
GPU's and the CPU oracle's final state) in numpy longdouble (x87 80-bit, 64-bit mantissa: a stand-in for exact arithmetic), following
reflector_ekf_slam.cc's formulas (:154-206 predict, :248-309 update, :311-364 augment) with the ORACLE's association lists (so that a last-place
difference cannot flip a match), and prints how far the GPU's and the oracle's means / covariances lie from it.  DIFF model, no map, no pose
observation (what the dumped seed uses).  CPU only (the oracle is the checker here, as in tests/):   python scripts/ld_witness.py file.npz"""
import sys
sys.path.insert(0, ".")
import numpy as np
from tests.helpers import make_oracle, norm_match
LD = np.longdouble
D = np.load(sys.argv[1])
assert int(D["model"]) == 0
lin, ang, obsv = float(D["lin"]), float(D["ang"]), float(D["obsv"])
o = make_oracle(0, float(D["init_time"]), D["init_pose"], lin, ang, obsv)
mu = np.array(D["init_pose"], LD); P = np.zeros((3, 3), LD); tm = LD(float(D["init_time"])); vt = np.zeros(3, LD)


def predict(dt):
    global mu, P
    v, w = vt[0], vt[2]
    dth = w * dt; half = mu[2] + dth / 2
    sh, ch = np.sin(half), np.cos(half)
    d = np.array([v * dt * ch, v * dt * sh, dth], LD)
    a, b = -v * dt * sh, v * dt * ch
    Gu = np.zeros((3, 2), LD); Gu[0] = [dt * ch, -v * dt * dt * sh / 2]; Gu[1] = [dt * sh, v * dt * dt * ch / 2]; Gu[2] = [0, dt]
    V = Gu @ np.diag(np.array([lin, ang], LD)) @ Gu.T
    P[0, :] = P[0, :] + a * P[2, :]; P[1, :] = P[1, :] + b * P[2, :]
    P[:, 0] = P[:, 0] + a * P[:, 2]; P[:, 1] = P[:, 1] + b * P[:, 2]
    P[:3, :3] += V
    mu[:3] += d
    mu[2] = np.arctan2(np.sin(mu[2]), np.cos(mu[2]))


def inv_ld(S):
    m = S.shape[0]; A = np.concatenate([S.copy(), np.eye(m, dtype=LD)], 1)
    for k in range(m):
        p = k + int(np.argmax(np.abs(A[k:, k]))); A[[k, p]] = A[[p, k]]
        A[k] = A[k] / A[k, k]
        for i in range(m):
            if i != k: A[i] = A[i] - A[i, k] * A[k]
    return A[:, m:]


for e in range(D["kind"].shape[0]):
    t = float(D["t"][e])
    if D["kind"][e] == 0:
        od = D["odom"][e]
        o.handle_odometry(t, *od)
        if LD(t) >= tm:                                                    # cc:211-212
            vt[:] = od; predict(LD(t) - tm); tm = LD(t)
        continue
    ob = D["obs"][D["obs_off"][e]: D["obs_off"][e + 1]]
    o.handle_observation(t, ob)
    sp, mp, nw = norm_match(o.last_match())
    predict(LD(t) - tm); tm = LD(t)
    if ob.shape[0] == 0: continue
    n = mu.shape[0]
    if sp.shape[0] > 0:
        m = 2 * sp.shape[0]
        H = np.zeros((m, n), LD); dz = np.zeros(m, LD)
        c, s = np.cos(mu[2]), np.sin(mu[2])
        for i, (lid, gid) in enumerate(sp):
            dx, dy = mu[3 + 2 * gid] - mu[0], mu[4 + 2 * gid] - mu[1]
            H[2 * i, :3] = [-c, -s, -dx * s + dy * c]; H[2 * i + 1, :3] = [s, -c, -dx * c - dy * s]
            H[2 * i, 3 + 2 * gid: 5 + 2 * gid] = [c, s]; H[2 * i + 1, 3 + 2 * gid: 5 + 2 * gid] = [-s, c]
            dz[2 * i: 2 * i + 2] = np.array([LD(ob[lid, 0]), LD(ob[lid, 1])]) - np.array([dx * c + dy * s, -dx * s + dy * c], LD)
        S = H @ P @ H.T + np.eye(m, dtype=LD) * LD(obsv)
        Kt = P @ H.T @ inv_ld(S)
        mu = mu + Kt @ dz
        mu[2] = np.arctan2(np.sin(mu[2]), np.cos(mu[2]))
        P = P - Kt @ H @ P
    if nw.shape[0] > 0:
        N2 = nw.shape[0]; s, c = np.sin(mu[2]), np.cos(mu[2])
        Gp = np.zeros((2 * N2, 3), LD); newmu = np.zeros(2 * N2, LD)
        for i, lid in enumerate(nw):
            rx, ry = LD(ob[lid, 0]), LD(ob[lid, 1])
            newmu[2 * i: 2 * i + 2] = [LD(np.float32(rx * c - ry * s + mu[0])), LD(np.float32(rx * s + ry * c + mu[1]))]      # cc:327-342
            Gp[2 * i] = [1, 0, -rx * s - ry * c]; Gp[2 * i + 1] = [0, 1, rx * c - ry * s]
        Gz = np.tile(np.array([[c, -s], [s, c]], LD), (N2, 1))
        Smm = Gp @ P[:3, :3] @ Gp.T + Gz @ (np.eye(2, dtype=LD) * LD(obsv)) @ Gz.T                                            # cc:354 (Q7)
        Smx = Gp @ P[:3, :]
        P = np.block([[P, Smx.T], [Smx, Smm]])
        mu = np.concatenate([mu, newmu])
mo, Po = o.state()
assert np.array_equal(mo, D["oracle_mu"]), "the replayed oracle differs from the dumped one"
g, Pg = D["gpu_mu"].astype(LD), D["gpu_P"].astype(LD)
print("n =", mu.shape[0], " events:", int(D["kind"].shape[0]))
print("max |oracle - longdouble|   mean %.3e   covariance %.3e" % (float(np.abs(mo.astype(LD) - mu).max()), float(np.abs(Po.astype(LD) - P).max())))
print("max |GPU    - longdouble|   mean %.3e   covariance %.3e" % (float(np.abs(g - mu).max()), float(np.abs(Pg - P).max())))
print("max |GPU    - oracle|       mean %.3e   covariance %.3e" % (float(np.abs(D["gpu_mu"] - mo).max()), float(np.abs(D["gpu_P"] - Po).max())))
print("asymmetry max |P - P^T|:  longdouble %.3e   oracle %.3e   GPU %.3e" % (float(np.abs(P - P.T).max()), float(np.abs(Po - Po.T).max()), float(np.abs(D["gpu_P"] - D["gpu_P"].T).max())))
