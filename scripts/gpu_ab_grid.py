"""Scan-to-pose latency (HandleObservationMessage + GetPose, device idle when the scan arrives) and the read-back rate with the match grid
on and off (rekf_debug_set_grid), one process, C3 full filter and the wrapper default.  GPU box: python scripts/gpu_ab_grid.py"""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
cfg = getattr(synth, sys.argv[1] if len(sys.argv) > 1 else "C3")
sess = synth.make_session(cfg)
scans = synth.steady_state_scans(sess, 1300)
for grid in (False, True, False, True):
    g = ReflectorEKFSLAM(S.options_for(sess)) if len(sys.argv) < 3 else ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
    g.debug_set_grid(grid)
    S.replay(sess, g); g.sync()
    idle = np.zeros(300); call = np.zeros(300)
    for k, (t, ob) in enumerate(scans[:300]):
        g.sync()
        t0 = time.perf_counter(); g.handle_observation(t, ob); t1 = time.perf_counter(); g.pose(); idle[k] = 1e6 * (time.perf_counter() - t0); call[k] = 1e6 * (t1 - t0)
    t0 = time.perf_counter()
    for t, ob in scans[300:1300]:
        g.handle_observation(t, ob); g.pose()
    rb = 1000 / (time.perf_counter() - t0)
    print(json.dumps({"cfg": cfg.name, "grid": grid, "idle_median_us": round(float(np.median(idle)), 2), "idle_p99_us": round(float(np.percentile(idle, 99)), 2), "call_median_us": round(float(np.median(call)), 2),
                      "readback_updates_per_s": round(rb, 1)}), flush=True)
    g.close()
