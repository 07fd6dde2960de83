"""A/B timing of builds of librekf.so in ONE GPU session (same box, same clocks), interleaved repetitions:
    python scripts/gpu_ab.py [--reps 3] path/to/A.so path/to/B.so ...
Per variant and repetition: us per update of the C3 steady state (1000 updates, host clock around enqueue + sync) and the
back-to-back k_downdate2 time (rekf_debug_time_kernel, 400 launches between one event pair).  Prints the medians.  REKF_AB_CFG=C2 (or C4) in the environment picks another BASELINE configuration."""
import subprocess, sys, os, statistics
BUILD = r'''
import sys
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
cfg = getattr(synth, __import__('os').environ.get('REKF_AB_CFG', 'C3'))
sess = synth.make_session(cfg)
g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
S.replay(sess, g); g.sync()
st = g.GetState()
np.savez("/tmp/c3_state.npz", t=st.time, mu=st.mu, sigma=st.sigma)
'''
CHILD = r'''
import sys, time
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import _lib
path = sys.argv[1]
_lib.lib_path = lambda name, _p=path: _p if name == "librekf.so" else __import__("os").path.join(_lib._HERE, name)
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
cfg = getattr(synth, __import__('os').environ.get('REKF_AB_CFG', 'C3'))
sess = synth.make_session(cfg)
z = np.load("/tmp/c3_state.npz")
g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
g.set_state(float(z["t"]), z["mu"], z["sigma"], (0.0, 0.0, 0.0))
scans = synth.steady_state_scans(sess, 1200)
for t, ob in scans[:100]:
    g.handle_observation(t, ob)
g.sync_code()
t0 = time.perf_counter()
for t, ob in scans[100:1100]:
    g.handle_observation(t, ob)
t_enq = (time.perf_counter() - t0) / 1000
g.sync_code()
dt = (time.perf_counter() - t0) / 1000
dd = g.time_kernel("downdate", reps=400)
print(f"RES {1e6 * dt:.3f} {dd:.3f} {1e6 * t_enq:.3f}")
'''
args = sys.argv[1:]
reps = 3
if args and args[0] == "--reps":
    reps = int(args[1]); args = args[2:]
subprocess.run([sys.executable, "-c", BUILD], check=True)
res = {p: [] for p in args}
for r in range(reps):
    for p in args:
        out = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(p)], capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("RES")]
        if not line:
            print(p, "FAILED", out.stderr[-600:]); continue
        _, a, b, c = line[-1].split()
        res[p].append((float(a), float(b), float(c)))
for p in args:
    if res[p]:
        print(f"{os.path.basename(p):28s} us/update median {statistics.median(x[0] for x in res[p]):7.2f}  (" + " ".join(f"{x[0]:.2f}" for x in res[p]) +
              f")   k_downdate2 back-to-back median {statistics.median(x[1] for x in res[p]):6.2f}  (" + " ".join(f"{x[1]:.2f}" for x in res[p]) + f")   host enqueue alone {statistics.median(x[2] for x in res[p]):.2f} us/update")
