"""k_mid's in-kernel timeline in the grid-match form (workgroup 1, s_memtime marks) from a -DREKF_DEBUG_TIMING -DREKF_DEBUG_GRID build:
    python scripts/gpu_dbg_grid_marks.py scripts/probe/librekf_dbg.so [C3]"""
import sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import _lib
path = sys.argv[1]
_lib.lib_path = lambda name, _p=path: _p if name == "librekf.so" else __import__("os").path.join(_lib._HERE, name)
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
cfg = getattr(synth, sys.argv[2] if len(sys.argv) > 2 else "C3")
sess = synth.make_session(cfg)
L = _lib.rekf(); L.rekf_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
scans = synth.steady_state_scans(sess, 40)
for grid in (True, False):
    g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
    g.debug_set_grid(grid)
    S.replay(sess, g); g.sync()
    for k, (t, ob) in enumerate(scans[:8]):
        g.handle_observation(t, ob); g.pose()
        if k >= 5:
            out = (C.c_longlong * 32)(); L.rekf_debug_counters(g._h, out)
            o = list(out)
            ghz = o[6] / max(o[5], 1) * 0.1
            print("grid", grid, "kernel %.2f us @ %.2f GHz marks(us):" % (o[5] * 0.01, ghz), [round(x / ghz / 1e3, 2) for x in o[8:8 + o[7]]])
    g.close()
