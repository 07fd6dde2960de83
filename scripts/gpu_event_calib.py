import sys
sys.path.insert(0, ".")
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
cfg = synth.C3
sess = synth.make_session(cfg)
g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks)
S.replay(sess, g); g.sync()
scans = synth.steady_state_scans(sess, 1300)
for only in (None, ["downdate"], ["downdate", "empty"], ["empty"], ["solve"], ["gain"]):
    g.profile_reset(); g.profile(True, only)
    for t, ob in scans[:400] if only is None else scans[400:800]:
        g.handle_observation(t, ob)
    g.profile(False)
    pr = g.profile_read()
    print(only, {k: round(v[0] / v[1], 2) for k, v in pr.items() if v[1]})
    scans = scans[400:] + scans[:400]
    sess_t = scans[-1][0]
