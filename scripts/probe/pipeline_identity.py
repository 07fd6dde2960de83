"""Which knob breaks bit-identity of the pose block after a few scans on a full filter: REKF_SPEC x REKF_SCAN_LAUNCH x (sync after every scan)."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from reflector_ekf_slam_amd import ReflectorEKFSLAM, synth
from reflector_ekf_slam_amd import session as S
L = int(os.environ.get("DBG_L", "140")); OBS = int(os.environ.get("DBG_OBS", "24")); NS = int(os.environ.get("DBG_NS", "6"))
cfg = synth.SessionConfig("r5_api", L, OBS, synth.DIFF, seed=5711, speed=1.4, row_spacing=6.0)
sess = synth.make_session(cfg)
scans = synth.steady_state_scans(sess, NS)
res = {}
for spec, fast, sync_each in itertools.product([0, 1], [0, 1], [0, 1]):
    os.environ["REKF_SPEC"] = str(spec); os.environ["REKF_SCAN_LAUNCH"] = str(fast)
    g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
    S.replay(sess, g); g.sync_code()
    for t, ob in scans:
        g.handle_observation(t, ob)
        if sync_each: g.sync_code()
    st = g.GetState()
    res[(spec, fast, sync_each)] = (st.mu.copy(), st.sigma.copy())
    g.close()
base = res[(0, 0, 1)]
for k, v in res.items():
    dm = np.max(np.abs(v[0] - base[0])); ds = np.abs(v[1] - base[1]); ij = np.unravel_index(np.argmax(ds), ds.shape)
    print(f"spec={k[0]} scan_launch={k[1]} sync_each={k[2]}: max|dmu|={dm:.3e} max|dSigma|={ds.max():.3e} at {ij}  differing entries {int((ds > 0).sum())}")
