"""Replays the C3 session and prints max |P - P^T| every ten scans: exactly 0 with the mirrored downdate (the product path);
with REKF_DD_FULL=1 (every tile computed, the reference's `sigma - K H sigma` as it stands) the antisymmetric part grows,
8e-15 -> 2.4e-14 over 540 steady-state scans at C3.  GPU box: python scripts/gpu_dbg_symmetry.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import ReflectorEKFSLAM, synth, session as S
cfg = synth.C3
sess = synth.make_session(cfg)
g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
state = {"last": None}
def on_scan(e, k):
    try:
        g.sync()
    except Exception as ex:
        print("scan", k, "ERR", ex); raise
    if k % 10 == 0 or k < 40:
        st = g.GetState()
        P = st.sigma; n = P.shape[0]
        asym = float(np.abs(P - P.T).max())
        T = n // 64
        A = np.abs(P - P.T)
        worst = np.unravel_index(np.argmax(A), A.shape)
        print(f"scan {k} n {n} T {T} rem {n%64} asym {asym:.3e} at {worst} diagmin {float(np.diag(P).min()):.3e} sum {float(np.abs(P).sum()):.6e}", flush=True)
try:
    S.replay(sess, g, on_scan=on_scan)
except Exception as ex:
    print("stopped:", ex)
