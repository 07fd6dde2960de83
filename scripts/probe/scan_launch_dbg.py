"""One launch per scan (REKF_SCAN_LAUNCH=1) against the two-launch chain (=0): first scan count at which the states differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from reflector_ekf_slam_amd import ReflectorEKFSLAM, synth
from reflector_ekf_slam_amd import session as S

L = int(os.environ.get("DBG_L", "100")); OBS = int(os.environ.get("DBG_OBS", "14")); EXCL = os.environ.get("DBG_EXCL", "0") == "1"
cfg = synth.SessionConfig("dbg", L, OBS, synth.DIFF, seed=5500 + L, speed=1.4, row_spacing=6.0)
sess = synth.make_session(cfg)
scans = synth.steady_state_scans(sess, int(os.environ.get("DBG_SCANS", "12")))

def run(fast, N):
    os.environ["REKF_SCAN_LAUNCH"] = "1" if fast else "0"
    g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
    g.set_exclusive(EXCL)
    S.replay(sess, g)
    g.GetState()
    for t, ob in scans[:N]:
        g.handle_observation(t, ob)
    st = g.GetState()
    fl = g.flags()
    g.close()
    return st, fl

for N in [int(x) for x in os.environ.get("DBG_N", "1,2,3,4,6,10").split(",")]:
    a, fa = run(False, N)
    b, fb = run(True, N)
    dmu = np.abs(a.mu - b.mu).max(); dS = np.abs(a.sigma - b.sigma)
    i, j = np.unravel_index(np.argmax(dS), dS.shape)
    print(f"N={N}: flags {fa} {fb}  max|dmu|={dmu:.3e}  max|dSigma|={dS.max():.3e} at ({i},{j})  n={a.mu.shape[0]}")
    if dS.max() > 0:
        bad = np.argwhere(dS > 1e-13)
        print("   differing entries:", len(bad), "rows", sorted(set(bad[:, 0]))[:20], "cols", sorted(set(bad[:, 1]))[:20])
