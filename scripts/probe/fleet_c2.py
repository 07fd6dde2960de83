"""Aggregate updates/s of S independent small-state sessions (BASELINE config 2: n = 259, 16 observations) sharing one GPU, one HIP stream each:
fed round-robin by ONE host thread, and by T host threads (ctypes releases the GIL inside the C ABI).  python scripts/probe/fleet_c2.py"""
import sys, time, threading
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
cfg = synth.C2
sess = synth.make_session(cfg)
steps = 600
scans = synth.steady_state_scans(sess, 100 + steps)
def make(ns):
    hs = []
    for _ in range(ns):
        g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
        S.replay(sess, g); g.sync(); hs.append(g)
    for t, ob in scans[:100]:
        for g in hs: g.handle_observation(t, ob)
    for g in hs: g.sync()
    return hs
def feed(hs):
    for t, ob in scans[100:]:
        for g in hs: g.handle_observation(t, ob)
    for g in hs: g.sync()
for ns, nt in [(1, 1), (2, 1), (4, 1), (8, 1), (16, 1), (8, 2), (8, 4), (16, 4), (16, 8), (32, 8)]:
    hs = make(ns)
    groups = [hs[i::nt] for i in range(nt)]
    t0 = time.perf_counter()
    if nt == 1: feed(hs)
    else:
        th = [threading.Thread(target=feed, args=(gr,)) for gr in groups]
        [x.start() for x in th]; [x.join() for x in th]
    dt = time.perf_counter() - t0
    ref = hs[0].mu()
    same = all(np.array_equal(g.mu(), ref) for g in hs[1:])
    print("%2d sessions, %d host thread(s): %7.0f updates/s aggregate (%.2f us per update per session), bit-identical %s, flags %s" % (ns, nt, ns * steps / dt, 1e6 * dt / steps, same, [g.flags() for g in hs[:2]]))
    for g in hs: g.close()
