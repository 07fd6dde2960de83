"""Entry / exit wall clock (100 MHz) of every workgroup of k_downdate2, inside the update chain, from a -DREKF_DEBUG_ENTRY build
(release register footprint; the exit stamp waits for the wave's own stores):
    python scripts/gpu_dbg_entry.py path/to/variant.so ...
Per variant, medians over 20 C3 updates, in us after the earliest entry: entry spread, exit by number of tiles / class, last exit."""
import subprocess, sys, os
CHILD = r'''
import sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import _lib
path = sys.argv[1]
_lib.lib_path = lambda name, _p=path: _p if name == "librekf.so" else __import__("os").path.join(_lib._HERE, name)
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
cfg = synth.C3
sess = synth.make_session(cfg)
g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
S.replay(sess, g); g.sync()
L = _lib.rekf()
L.rekf_debug_dd_times.argtypes = [C.c_void_p, C.c_int]
scans = synth.steady_state_scans(sess, 100)
G = 256
rows = []
hip = C.CDLL("libamdhip64.so")
it = iter(scans[10:])
for rep in range(20):
    for _ in range(3):                       # scan after scan: the second and third run as k_dd_front + k_mid
        t, ob = next(it); g.handle_observation(t, ob)
    hip.hipDeviceSynchronize()               # (not g.sync(): that would enqueue the held-back downdate as k_downdate2 and overwrite the stamps)
    buf = (C.c_longlong * (2 * G))()
    L.rekf_debug_dd_times(buf, G)
    a = np.array(list(buf), float).reshape(G, 2) * 0.01
    rows.append(a - a[:, 0].min())
m = np.median(np.array(rows), axis=0)
dur = m[:, 1]
GD = 224                       # k_dd_front at C3: 224 downdate workgroups, then 32 front-end workgroups (k_downdate2 alone: GD = G)
if "--alone" in sys.argv: GD = G
front = dur[GD:]
dur = dur[:GD]; m = m[:GD]
w = np.array([(b & 7) * (GD >> 3) + (b >> 3) for b in range(GD)])
T = 32
A = w < T
if front.size: print("front-end workgroups: exits median %.2f max %.2f us" % (np.median(front), front.max()))
order = np.argsort(dur)
print(path.split("/")[-1], f"entries {m[:, 0].min():.2f}..{m[:, 0].max():.2f}; exits: class A median {np.median(dur[A]):.2f} max {dur[A].max():.2f} (block 0: {dur[0]:.2f}); "
      f"class B quartiles {np.percentile(dur[~A], 25):.2f} {np.percentile(dur[~A], 50):.2f} {np.percentile(dur[~A], 75):.2f} max {dur[~A].max():.2f}; "
      f"last exit {dur.max():.2f}; sum of bodies {np.sum(m[:, 1] - m[:, 0]):.0f} us; latest 6 (w, exit): " + " ".join(f"({w[i]},{dur[i]:.2f})" for i in order[-6:]))
'''
for p in sys.argv[1:]:
    r = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(p)], capture_output=True, text=True)
    print("\n".join((r.stdout.strip().splitlines() or ["(no output)"])[-2:]))
    if r.returncode != 0:
        print(r.stderr[-800:])
