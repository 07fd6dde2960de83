import sys, ctypes as C, os
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import _lib
path = sys.argv[1]
_lib.lib_path = lambda name, _p=path: _p if name == "librekf.so" else os.path.join(_lib._HERE, name)
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
cfg = getattr(synth, sys.argv[2] if len(sys.argv) > 2 else "C2")
sess = synth.make_session(cfg)
g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
S.replay(sess, g); g.sync()
L = _lib.rekf(); L.rekf_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
scans = synth.steady_state_scans(sess, 40)
for k, (t, ob) in enumerate(scans[:8]):
    g.handle_observation(t, ob)
    if k == 6:
        out = (C.c_longlong * 32)(); L.rekf_debug_counters(g._h, out)
        o = list(out)
        ghz = o[6] / max(o[5], 1) * 0.1
        print("ONE_LAUNCH=%s: downdate body of workgroup 0: %.2f us @ %.2f GHz, marks(us):" % (os.environ.get("REKF_ONE_LAUNCH", "1"), o[5] * 0.01, ghz), [round(x / ghz / 1e3, 2) for x in o[8:8 + o[7]]])
