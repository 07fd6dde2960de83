"""Multi-session aggregate with ONE HOST THREAD PER SESSION (ctypes releases the GIL inside a call) against the bench's single feeding thread.
GPU box: python scripts/gpu_ab_multi_threads.py [steps]"""
import json, os, sys, threading, time
sys.path.insert(0, ".")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
cfg = synth.C3
sess = synth.make_session(cfg)
scans = synth.steady_state_scans(sess, steps + 100)
for mode in ("default", "fixed"):
    for nsess in (2, 4):
        gs = []
        for _ in range(nsess):
            g = ReflectorEKFSLAM(S.options_for(sess)) if mode == "default" else ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
            S.replay(sess, g); g.sync(); gs.append(g)
        for t, ob in scans[:100]:
            for g in gs: g.handle_observation(t, ob)
        for g in gs: g.sync()
        bar = threading.Barrier(nsess + 1)
        def work(g):
            bar.wait()
            for t, ob in scans[100:]: g.handle_observation(t, ob)
            g.sync()
        th = [threading.Thread(target=work, args=(g,)) for g in gs]
        [x.start() for x in th]
        bar.wait(); t0 = time.perf_counter()
        [x.join() for x in th]
        dt = time.perf_counter() - t0
        print(json.dumps({"mode": mode, "sessions": nsess, "threads": nsess, "updates_per_s_aggregate": round(nsess * steps / dt, 1)}), flush=True)
        for g in gs: g.close()
