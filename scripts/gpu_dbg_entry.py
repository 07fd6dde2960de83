"""Entry / exit wall-clock times of eight k_downdate2 workgroups (blocks 0, 128, 255, 256, 300, 400, 495, 511) relative to block 0,
from a -DREKF_DEBUG_ENTRY build (release register footprint: the -DREKF_DEBUG_TIMING build holds its marks in 48 VGPRs and
loses the second workgroup per CU).  GPU box: make -C reflector_ekf_slam_amd/csrc -B ../librekf.so HIPFLAGS="... -DREKF_DEBUG_ENTRY";
[REKF_DD_SB=1] python scripts/gpu_dbg_entry.py"""
import sys, ctypes as C
sys.path.insert(0, ".")
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM, _lib
cfg = synth.C3
sess = synth.make_session(cfg)
g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks)
S.replay(sess, g); g.sync()
L = _lib.rekf(); L.rekf_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
for rep in range(3):
    for t, ob in synth.steady_state_scans(sess, 20)[rep * 5:rep * 5 + 5]:
        g.handle_observation(t, ob)
    out = (C.c_longlong * 32)(); L.rekf_debug_counters(g._h, out)
    o = list(out)
    print("entry us:", [round((x - o[8]) * 0.01, 2) for x in o[8:16]], " exit us:", [round((x - o[8]) * 0.01, 2) for x in o[16:24]])
