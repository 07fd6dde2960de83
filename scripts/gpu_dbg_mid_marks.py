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
for k, (t, ob) in enumerate(scans[:12]):
    g.handle_observation(t, ob)
    if k >= 8:
        out = (C.c_longlong * 32)(); L.rekf_debug_counters(g._h, out)
        o = list(out)
        ghz = o[6] / max(o[5], 1) * 0.1
        print("kernel %.2f us @ %.2f GHz marks(us):" % (o[5] * 0.01, ghz), [round(x / ghz / 1e3, 2) for x in o[8:8 + o[7]]])
        # wall clock (100 MHz) of the LAST launch's roles, relative to mid workgroup 1's entry (dbg[4])
        e1 = o[4]
        print("   rel. to mid wg 1's entry: wg 1 exit %.2f | wg 0 exit %.2f | last mid wg: entry %.2f | last exit of any mid wg %.2f | downdate role: first start %.2f, last end %.2f us"
              % tuple((o[k] - e1) * 0.01 for k in (31, 29, 25, 30, 26, 27)))
