import sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM, _lib
cfg = synth.C3
sess = synth.make_session(cfg)
g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks)
S.replay(sess, g); g.sync()
L = _lib.rekf(); L.rekf_debug_counters.argtypes=[C.c_void_p, C.POINTER(C.c_longlong)]
for t, ob in synth.steady_state_scans(sess, 20):
    g.handle_observation(t, ob)
out = (C.c_longlong*32)(); L.rekf_debug_counters(g._h, out)
print("dbg", list(out)[:8], "tail", list(out)[24:32], "\ndowndate marks", list(out)[8:], "shader clocks per wallclock tick(100MHz):", out[0]/max(out[1],1), "-> GHz", out[0]/max(out[1],1)*0.1)
