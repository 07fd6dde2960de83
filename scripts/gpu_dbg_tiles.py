"""Which 64 x 64 tiles of P differ from the oracle after one update?  (debugging aid for k_downdate2's tile schedule)
usage: python scripts/gpu_dbg_tiles.py [L] [obs]"""
import sys
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import synth
from tests.helpers import make_gpu, make_oracle, norm_match
L = int(sys.argv[1]) if len(sys.argv) > 1 else 96
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(L)
n = 3 + 2 * L
side = int(np.ceil(np.sqrt(L)))
lm = np.array([(2.5 * (k % side), 2.5 * (k // side)) for k in range(L)], dtype=np.float64)
pose = np.array([2.5 * side / 2 + 0.3, 2.5 * side / 2 - 0.4, 0.3])
mu = np.concatenate([pose, (lm + rng.normal(0, 0.01, lm.shape)).ravel()])
A = rng.normal(size=(n, 40))
P = (A @ A.T) * 1e-5 + np.diag(rng.uniform(1e-4, 4e-4, n))
P = 0.5 * (P + P.T)
g = make_gpu(0, 0.0, pose, 0.0025, 0.0064, 0.0025, L)
o = make_oracle(0, 0.0, pose, 0.0025, 0.0064, 0.0025)
g.set_state(1.0, mu, P, (0.4, 0.0, 0.1))
o.set_state(1.0, mu, P, (0.4, 0.0, 0.1))
c, s_ = np.cos(pose[2]), np.sin(pose[2])
near = np.argsort(np.linalg.norm(lm - pose[:2], axis=1))[:K]
for k in range(2):
    rel = lm[near] - pose[:2]
    ob = np.stack([rel[:, 0] * c + rel[:, 1] * s_, -rel[:, 0] * s_ + rel[:, 1] * c], axis=1)
    ob = (ob + rng.normal(0, 0.01, ob.shape)).astype(np.float32)
    t = 1.0 + 0.05 * (k + 1)
    g.handle_observation(t, ob); o.handle_observation(t, ob)
    st = g.GetState(); mo, Po = o.state()
    D = np.abs(st.sigma - Po)
    T = (n + 63) // 64
    bad = [(i, j, float(D[64 * i:64 * i + 64, 64 * j:64 * j + 64].max())) for i in range(T) for j in range(i + 1)
           if D[64 * i:64 * i + 64, 64 * j:64 * j + 64].max() > 1e-11]
    print(f"n={n} T={T} scan {k}: max|dmu| {np.abs(st.mu - mo).max():.2e} max|dP| {D.max():.2e}; bad lower tiles (I, J, err): {bad[:12]} ({len(bad)} in all)")
