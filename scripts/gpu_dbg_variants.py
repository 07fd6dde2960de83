"""A/B timing of experimental builds of librekf.so (debug-timing builds: -DREKF_DEBUG_TIMING gives the in-kernel marks of
k_mid's workgroup 1; add -DREKF_DEBUG_DD2 for k_downdate2's).  Usage on the GPU box:
    python scripts/gpu_dbg_variants.py path/to/variantA.so path/to/variantB.so ...
The C3 state is built once with the release library; each variant (its own process) restores it and runs steady-state updates --
variants that deliberately compute wrong numbers (ablations) are read after their FIRST update, from a sane state."""
import subprocess, sys, os
BUILD = r'''
import sys
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
cfg = synth.C3
sess = synth.make_session(cfg)
g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
S.replay(sess, g); g.sync()
st = g.GetState()
np.savez("/tmp/c3_state.npz", t=st.time, mu=st.mu, sigma=st.sigma)
'''
CHILD = r'''
import sys, time, ctypes as C
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import _lib
path = sys.argv[1]
_lib.lib_path = lambda name, _p=path: _p if name == "librekf.so" else __import__("os").path.join(_lib._HERE, name)
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
cfg = synth.C3
sess = synth.make_session(cfg)
z = np.load("/tmp/c3_state.npz")
g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
g.set_state(float(z["t"]), z["mu"], z["sigma"], (0.0, 0.0, 0.0))
L = _lib.rekf(); L.rekf_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
scans = synth.steady_state_scans(sess, 1300)
res = []
for t, ob in scans[:3]:
    g.handle_observation(t, ob)
    out = (C.c_longlong * 32)(); L.rekf_debug_counters(g._h, out)
    o = list(out)
    ghz = o[6] / max(o[5], 1) * 0.1
    res.append((o[5] * 0.01, ghz, [round(x / ghz / 1e3, 2) for x in o[8:8 + o[7]]] if o[7] else [], (o[4] - o[3]) * 0.01))
g.set_state(float(z["t"]), z["mu"], z["sigma"], (0.0, 0.0, 0.0))
g.sync_code()
t0 = time.perf_counter()
for t, ob in scans[100:600]:
    g.handle_observation(t, ob)
g.sync_code()
dt = (time.perf_counter() - t0) / 500
print(f"{path.split('/')[-1]}: {1e6 * dt:.2f} us/update (meaningless for ablations); first update: kernel {res[0][0]:.2f} us @ {res[0][1]:.2f} GHz, marks {res[0][2]}; second: {res[1][2]}; entry of the recorded workgroup after workgroup 0's (k_downdate2 builds): {res[1][3]:.2f} us, body {res[1][0]:.2f} us")
'''
subprocess.run([sys.executable, "-c", BUILD], check=True)
for p in sys.argv[1:]:
    r = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(p)], capture_output=True, text=True)
    print((r.stdout.strip().splitlines() or ["(no output)"])[-1])
    if r.returncode != 0:
        print(r.stderr[-800:])
