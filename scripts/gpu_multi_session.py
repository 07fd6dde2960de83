"""Aggregate update rate of S independent filter sessions sharing ONE GPU (one handle = one HIP stream each, fed
round-robin by one host thread): the per-update chain is latency-bound, so sessions interleave.
GPU box: python scripts/gpu_multi_session.py [steps]"""
import sys, time, json
sys.path.insert(0, ".")
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
cfg = synth.C3
sess = synth.make_session(cfg)
scans = synth.steady_state_scans(sess, steps + 100)
for nsess in (1, 2, 3, 4, 6, 8):
    gs = []
    for _ in range(nsess):
        g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
        S.replay(sess, g); g.sync()
        gs.append(g)
    for t, ob in scans[:100]:
        for g in gs:
            g.handle_observation(t, ob)
    for g in gs:
        g.sync()
    t0 = time.perf_counter()
    for t, ob in scans[100:]:
        for g in gs:
            g.handle_observation(t, ob)
    t1 = time.perf_counter()
    for g in gs:
        g.sync()
    t2 = time.perf_counter()
    ref = gs[0].mu()
    same = all(float(abs(g.mu() - ref).max()) == 0.0 for g in gs)
    print(json.dumps({"sessions": nsess, "updates_per_s_aggregate": round(nsess * steps / (t2 - t0), 1),
                      "us_per_update_per_session": round((t2 - t0) / steps * 1e6, 2), "host_enqueue_us_per_update": round((t1 - t0) / steps / nsess * 1e6, 2),
                      "sessions_bit_identical": same}))
    for g in gs:
        g.close() if hasattr(g, "close") else None
    del gs
