"""Host enqueue rate against device rate for the steady-state update loop (is the chain GPU-bound or launch-bound?).
GPU box: [REKF_OVERLAP=0] python scripts/gpu_host_rate.py"""
import sys, time, json
sys.path.insert(0, ".")
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
cfg = synth.C3
sess = synth.make_session(cfg)
g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks)
S.replay(sess, g); g.sync()
scans = synth.steady_state_scans(sess, 2200)
for t, ob in scans[:200]:
    g.handle_observation(t, ob)
try:
    g.sync()
except Exception as e:
    print("sync:", e)
t0 = time.perf_counter()
for t, ob in scans[200:]:
    g.handle_observation(t, ob)
t1 = time.perf_counter()
try:
    g.sync()
except Exception as e:                                  # REKF_OVERLAP=2 (unsynchronised timing experiment) leaves garbage behind
    print("sync:", e)
t2 = time.perf_counter()
print(json.dumps({"steps": 2000, "host_enqueue_us_per_step": round((t1 - t0) / 2000 * 1e6, 2), "total_us_per_step": round((t2 - t0) / 2000 * 1e6, 2)}))
