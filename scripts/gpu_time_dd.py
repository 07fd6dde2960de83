"""Times k_downdate (and the other re-launchable kernels) in steady state at C3 via rekf_debug_time_kernel."""
import sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM, _lib
cfg = synth.C3
sess = synth.make_session(cfg)
g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks)
S.replay(sess, g); g.sync()
for t, ob in synth.steady_state_scans(sess, 5):
    g.handle_observation(t, ob)
g.sync()
L = _lib.rekf(); L.rekf_debug_time_kernel.argtypes=[C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
for name, k in (("solve",3),("gain",4),("downdate",5)):
    us = C.c_double(); rc = L.rekf_debug_time_kernel(g._h, k, 300, 0, C.byref(us))
    print(name, "rc", rc, "avg_us %.2f" % us.value)
for ab in (0, 512, 0, 8192, 0, 512, 0):
    us = C.c_double(); rc = L.rekf_debug_time_kernel(g._h, 5, 300, ab, C.byref(us))
    print("downdate ablate", ab, "avg_us %.2f" % us.value)
