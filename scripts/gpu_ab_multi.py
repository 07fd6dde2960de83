"""A/B of the multi-session aggregate (S sessions sharing one GPU, one host thread, round-robin) under an environment knob:
    python scripts/gpu_ab_multi.py [steps]        (REKF_* in the environment are read at rekf_create)
Prints one JSON line per (sessions, mode): mode "default" = the wrapper's defaults (growing), "fixed" = max_landmarks = L, auto_grow off."""
import json, os, sys, time
sys.path.insert(0, ".")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
cfg = synth.C3
sess = synth.make_session(cfg)
scans = synth.steady_state_scans(sess, steps + 100)
for mode in ("default", "fixed"):
    for nsess in (1, 4):
        gs = []
        for _ in range(nsess):
            g = ReflectorEKFSLAM(S.options_for(sess)) if mode == "default" else ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
            S.replay(sess, g); g.sync()
            gs.append(g)
        for t, ob in scans[:100]:
            for g in gs: g.handle_observation(t, ob)
        for g in gs: g.sync()
        t0 = time.perf_counter()
        for t, ob in scans[100:]:
            for g in gs: g.handle_observation(t, ob)
        for g in gs: g.sync()
        dt = time.perf_counter() - t0
        print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("REKF_")}, "mode": mode, "sessions": nsess,
                          "updates_per_s_aggregate": round(nsess * steps / dt, 1)}), flush=True)
        for g in gs: g.close()
