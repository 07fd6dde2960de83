"""k_mid's in-kernel timeline (workgroup 1, s_memtime marks) at a BASELINE configuration, from a -DREKF_DEBUG_TIMING build of librekf.so:
    hipcc ... -DREKF_DEBUG_TIMING ... -o scripts/probe/librekf_dbg.so      (flags as in csrc/Makefile)
    gpurun -- 'python scripts/gpu_dbg_mid_marks.py scripts/probe/librekf_dbg.so C2'
Marks (us from the kernel's entry): record in, compaction, gathers issued, W, S built | inverse | gain, stores."""
import sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import _lib
path = sys.argv[1]
_lib.lib_path = lambda name, _p=path: _p if name == "librekf.so" else __import__("os").path.join(_lib._HERE, name)
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
cfg = getattr(synth, sys.argv[2] if len(sys.argv) > 2 else "C2")
sess = synth.make_session(cfg)
g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
S.replay(sess, g); g.sync()
L = _lib.rekf(); L.rekf_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
scans = synth.steady_state_scans(sess, 40)
for k, (t, ob) in enumerate(scans[:6]):
    g.handle_observation(t, ob)
    if k >= 3:
        out = (C.c_longlong * 32)(); L.rekf_debug_counters(g._h, out)
        o = list(out)
        ghz = o[6] / max(o[5], 1) * 0.1
        print("kernel %.2f us @ %.2f GHz marks(us):" % (o[5] * 0.01, ghz), [round(x / ghz / 1e3, 2) for x in o[8:8 + o[7]]])
        if o[26]:                                     # the one-launch form: wall clock (100 MHz) of the roles, relative to the downdate role's entry
            t0 = o[26]
            print("   one launch: downdate role wg 0: body done %.2f, counted %.2f | front role wg 0: entry %.2f, exit %.2f | mid wg 1: entry %.2f, exit %.2f us"
                  % tuple((o[k] - t0) * 0.01 for k in (27, 28, 29, 30, 4, 31)))
