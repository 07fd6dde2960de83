"""Entry / exit wall clock of every workgroup of k_dd_front at C2 (15 downdate workgroups: 4 class A (diagonal tiles; workgroup 0 also publishes and
takes the pose block), 1 empty (T = 4 with strips -> 5 slots?), the rest class B with one tile each; then 16 front-end workgroups), -DREKF_DEBUG_ENTRY build."""
import sys, ctypes as C, os
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import _lib
path = os.path.abspath(sys.argv[1])
_lib.lib_path = lambda name, _p=path: _p if name == "librekf.so" else os.path.join(_lib._HERE, name)
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
cfg = synth.C2
sess = synth.make_session(cfg)
g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
S.replay(sess, g); g.sync()
L = _lib.rekf()
L.rekf_debug_dd_times.argtypes = [C.c_void_p, C.c_int]
scans = synth.steady_state_scans(sess, 100)
hip = C.CDLL("libamdhip64.so")
it = iter(scans[10:])
G = 40
rows = []
for rep in range(20):
    for _ in range(3):
        t, ob = next(it); g.handle_observation(t, ob)
    hip.hipDeviceSynchronize()
    buf = (C.c_longlong * (2 * G))()
    L.rekf_debug_dd_times(buf, G)
    a = np.array(list(buf), float).reshape(G, 2) * 0.01
    a[a[:, 0] == 0] = np.nan
    rows.append(a - np.nanmin(a[:, 0]))
m = np.nanmedian(np.array(rows), axis=0)
for b in range(G):
    if not np.isnan(m[b, 0]): print("wg %2d  entry %5.2f  exit %5.2f" % (b, m[b, 0], m[b, 1]))
