import os, sys, numpy as np, ctypes as C
sys.path.insert(0, ".")
from reflector_ekf_slam_amd import synth, ReflectorEKFSLAM, session as S
cfg = synth.SessionConfig("r4_one_launch", 100, 14, synth.DIFF, seed=4420, speed=1.4, row_spacing=6.0)
sess = synth.make_session(cfg)
scans = synth.steady_state_scans(sess, 240)
rng = np.random.default_rng(7)
scans = [(t, np.concatenate([ob, rng.uniform(-9.0, 9.0, (1, 2)).astype(np.float32)]) if k % 9 == 4 else ob) for k, (t, ob) in enumerate(scans)]
def run(one, N):
    if one: os.environ["REKF_ONE_LAUNCH"] = "1"
    else: os.environ.pop("REKF_ONE_LAUNCH", None)
    g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=int(os.environ.get("CAP", "8")), auto_grow=os.environ.get("CAP") is None)
    S.replay(sess, g)
    for k, (t, ob) in enumerate(scans[:N]):
        g.handle_observation(t, ob)
        if k % 53 == 52: g.pose()
    out = (C.c_longlong * 32)(); g._L.rekf_debug_counters(g._h, out)
    st = g.GetState()
    return st, int(out[24]), g.max_landmarks
for N in [int(x) for x in sys.argv[1].split(',')]:
    (a, ra, ca), (b, rb, cb) = run(False, N), run(True, N)
    print("N %d: n %d cap %d/%d roles %d, max|dmu| %.3e max|dP| %.3e" % (N, a.mu.shape[0], ca, cb, rb, np.abs(a.mu - b.mu).max(), np.abs(a.sigma - b.sigma).max()))
print("determinism:")
(a, _, _), (b, _, _) = run(False, 240), run(False, 240)
print(" two-launch twice: max|dmu| %.3e" % np.abs(a.mu - b.mu).max())
(c, r1, _), (d, r2, _) = run(True, 240), run(True, 240)
print(" one-launch twice: roles %d %d max|dmu| %.3e ; vs two-launch %.3e %.3e" % (r1, r2, np.abs(c.mu - d.mu).max(), np.abs(a.mu - c.mu).max(), np.abs(a.mu - d.mu).max()))
