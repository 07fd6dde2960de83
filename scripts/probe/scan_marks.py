"""In-kernel timeline of the one-launch-per-scan form (C3, pipelined): -DREKF_DEBUG_TIMING build (scripts/probe/librekf_dbg.so).
Mid role marks of mid workgroup 1 (us from its entry); the downdate role's first workgroup and the LAST exit of any downdate workgroup,
relative to mid workgroup 1's entry."""
import sys, os, ctypes as C
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import _lib
path = sys.argv[1]
_lib.lib_path = lambda name, _p=path: _p if name == "librekf.so" else __import__("os").path.join(_lib._HERE, name)
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
cfg = getattr(synth, sys.argv[2] if len(sys.argv) > 2 else "C3")
excl = len(sys.argv) > 3 and sys.argv[3] == "1"
sess = synth.make_session(cfg)
g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
g.set_exclusive(excl)
S.replay(sess, g); g.sync_code()
L = _lib.rekf(); L.rekf_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
scans = synth.steady_state_scans(sess, 60)
k = 0
for rep in range(3):
    for t, ob in scans[k:k + 8]: g.handle_observation(t, ob)
    k += 8
    out = (C.c_longlong * 32)(); L.rekf_debug_counters(g._h, out)
    o = list(out)
    ghz = o[6] / max(o[5], 1) * 0.1
    print("mid wg 1: %.2f us @ %.2f GHz marks(us):" % (o[5] * 0.01, ghz), [round(x / ghz / 1e3, 2) for x in o[8:8 + o[7]]])
    if o[28]:
        print("   launch timeline (us after the first workgroup's entry): mid wg 1 in %.2f out %.2f | last mid wg in %.2f | wg 0 out %.2f | last mid out %.2f | dd role first in %.2f, last out %.2f"
              % tuple((o[q] - o[28]) * 0.01 for q in (4, 31, 25, 29, 30, 26, 27)))
    if o[26]:
        t0 = o[4]
        print("   relative to mid wg 1's entry: its exit %.2f | first downdate wg: entry %.2f exit %.2f | LAST downdate wg exit %.2f us"
              % tuple((o[q] - t0) * 0.01 for q in (31, 26, 28, 27)))
