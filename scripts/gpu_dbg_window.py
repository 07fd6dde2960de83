"""Where a 20-update window (the driver's command: sync, 20 x HandleObservationMessage, sync) spends its time: host timestamps after every
call and after the closing sync, median over repetitions.  GPU box: python scripts/gpu_dbg_window.py [steps]"""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import _lib
if len(sys.argv) > 2:                               # another build of librekf.so (A/B of experimental builds)
    _lib.lib_path = lambda name, _p=sys.argv[2]: _p if name == "librekf.so" else __import__("os").path.join(_lib._HERE, name)
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = synth.C3
sess = synth.make_session(cfg)
scans = synth.steady_state_scans(sess, 60 * (steps + 5))
g = ReflectorEKFSLAM(S.options_for(sess))
S.replay(sess, g); g.sync()
reps = 50
T = np.zeros((reps, steps + 1))
it = iter(scans)
for r in range(reps):
    for _ in range(5):
        t, ob = next(it); g.handle_observation(t, ob)
    g.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        t, ob = next(it); g.handle_observation(t, ob)
        T[r, k] = time.perf_counter() - t0
    g.sync()
    T[r, steps] = time.perf_counter() - t0
med = np.median(T, axis=0) * 1e6
print("after each call (us):", [round(float(x), 1) for x in med[:steps]])
print("after sync:", round(float(med[steps]), 1), "us ->", round(steps / med[steps] * 1e6), "updates/s; deltas:", [round(float(x), 1) for x in np.diff(med)])
