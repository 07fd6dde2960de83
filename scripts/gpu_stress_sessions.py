"""Several filter handles on one GPU, driven round-robin by one host thread in the patterns that put waiting workgroups into a launch
(round 4): a pose read-back after every scan (the scan's front end runs inside k_mid's grid, the mid workgroups wait for it) and
growing filters enqueued without read-backs (the previous scan's augmentation runs inside k_mid, the other workgroups wait for it
when there is something to append) -- with the other handles' persistent downdates competing for the CUs.  Every handle must end
bit-identical to a handle that ran the same session alone.
GPU box: python scripts/gpu_stress_sessions.py [sessions] [L]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cfg = synth.SessionConfig(f"stress_L{L}", L, 24, synth.DIFF, seed=909, speed=1.5, row_spacing=6.0)
sess = synth.make_session(cfg)
events = [(int(sess.ev_type[e]), float(sess.ev_time[e]), e) for e in range(sess.n_events)]


def run(handles, readback):
    first = True
    for typ, t, e in events:
        if typ == synth.EV_ODOM:
            for g in handles:
                g.handle_odometry(t, *sess.odom[e])
            continue
        if first:
            first = False
            continue
        ob = sess.obs_of(e)
        for g in handles:
            g.handle_observation(t, ob)
        if readback:
            for g in handles:
                g.pose()
    for g in handles:
        assert g.sync_code() == 0
    return [g.GetState() for g in handles]


for readback in (True, False):
    ref = run([ReflectorEKFSLAM(S.options_for(sess), max_landmarks=8)], readback)[0]          # grows 8 -> ... on the way (auto-grow)
    t0 = time.time()
    sts = run([ReflectorEKFSLAM(S.options_for(sess), max_landmarks=8) for _ in range(ns)], readback)
    same = all(np.array_equal(st.mu, ref.mu) and np.array_equal(st.sigma, ref.sigma) for st in sts)
    worst = max(float(np.abs(st.mu - ref.mu).max()) if st.mu.shape == ref.mu.shape else float("inf") for st in sts)
    print(f"read-back after every scan: {readback}; {ns} sessions, n = {ref.mu.shape[0]}: bit-identical to the lone session: {same} (max |mu - lone| = {worst:.3e}); {time.time() - t0:.1f} s", flush=True)
    assert worst < 1e-9
