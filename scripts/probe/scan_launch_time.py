"""us per update at C3 (pipelined, full filter) for the launch forms: REKF_SCAN_LAUNCH x exclusive."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from reflector_ekf_slam_amd import ReflectorEKFSLAM, synth
from reflector_ekf_slam_amd import session as S

L = int(os.environ.get("DBG_L", "1024")); OBS = int(os.environ.get("DBG_OBS", "32"))
cfg = synth.SessionConfig("tm", L, OBS, synth.DIFF, seed=20210331, speed=1.4, row_spacing=6.0)
sess = synth.make_session(cfg)
scans = synth.steady_state_scans(sess, 2300)
CFGS = [tuple(int(c) for c in x) for x in os.environ.get("DBG_CFGS", "00,10,11,01,11,10").split(",")]
for fast, excl in CFGS:
    os.environ["REKF_SCAN_LAUNCH"] = str(fast)
    g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
    g.set_exclusive(bool(excl))
    S.replay(sess, g)
    g.sync()
    for t, ob in scans[:300]: g.handle_observation(t, ob)
    g.sync()
    t0 = time.perf_counter()
    for t, ob in scans[300:2300]: g.handle_observation(t, ob)
    g.sync()
    dt = time.perf_counter() - t0
    import ctypes as C
    out = (C.c_longlong * 32)(); g._L.rekf_debug_counters(g._h, out)
    print(f"scan_launch={fast} exclusive={excl}: {dt / 2000 * 1e6:.2f} us/update  ({2000 / dt:.0f} updates/s) flags={g.flags()}  corrected scans {out[22]}, of which computed in k_mid {out[23]}; speculative records {out[20]}, with re-matched observations {out[21]}")
    g.close()
