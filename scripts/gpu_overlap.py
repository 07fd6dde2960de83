"""Can k_downdate run under the next update's latency chain?  (rekf_debug_overlap)  GPU box: python scripts/gpu_overlap.py"""
import sys, ctypes as C, json
sys.path.insert(0, ".")
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM, _lib
cfg = synth.C3
sess = synth.make_session(cfg)
g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks)
S.replay(sess, g); g.sync()
for t, ob in synth.steady_state_scans(sess, 5):
    g.handle_observation(t, ob)
g.sync()
L = _lib.rekf()
L.rekf_debug_overlap.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
out = (C.c_double * 3)()
for reps in (200, 1000):
    rc = L.rekf_debug_overlap(g._h, reps, out)
    print(json.dumps({"rc": rc, "reps": reps, "downdate_alone_us": round(out[0], 2), "solve_gain_alone_us": round(out[1], 2),
                      "both_streams_us": round(out[2], 2), "sum_us": round(out[0] + out[1], 2)}))
