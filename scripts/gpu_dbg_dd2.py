"""In-kernel timeline of k_downdate2 (workgroup 0) from a -DREKF_DEBUG_TIMING build of librekf.so, C3 steady state.
Run on the GPU box:  make -C reflector_ekf_slam_amd/csrc HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -DREKF_DEBUG_TIMING -DREKF_DEBUG_DD2" (timeline of k_downdate2) or without -DREKF_DEBUG_DD2 (timeline of k_mid, workgroup 1)  first."""
import sys, ctypes as C
sys.path.insert(0, ".")
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM, _lib
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = getattr(synth, name)
sess = synth.make_session(cfg)
g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
S.replay(sess, g); g.sync()
L = _lib.rekf(); L.rekf_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
for rep in range(3):
    for t, ob in synth.steady_state_scans(sess, 20)[rep * 5:rep * 5 + 5]:
        g.handle_observation(t, ob)
    out = (C.c_longlong * 32)(); L.rekf_debug_counters(g._h, out)
    o = list(out)
    ghz = o[6] / max(o[5], 1) * 0.1
    print(f"total {o[6]} cycles = {o[5] * 10} ns wall -> {ghz:.2f} GHz; entered {(o[4] - o[3]) * 0.01:.2f} us after block 0; marks({o[7]}):", [round(x / ghz / 1e3, 2) for x in o[8:8 + o[7]]], "us")
