"""Where a 20-update window's time goes (the driver's bench command times 20 updates between two syncs): per-call host time of each
handle_observation, the closing sync, over many repetitions; median of each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from reflector_ekf_slam_amd import ReflectorEKFSLAM, synth
from reflector_ekf_slam_amd import session as S

cfg = synth.SessionConfig("tm", 1024, 32, synth.DIFF, seed=20210331, speed=1.4, row_spacing=6.0)
sess = synth.make_session(cfg)
K = int(os.environ.get("DBG_K", "20")); REPS = 60
scans = synth.steady_state_scans(sess, 300 + REPS * K)
g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
S.replay(sess, g); g.sync()
for t, ob in scans[:300]: g.handle_observation(t, ob)
g.sync()
calls = np.zeros((REPS, K)); fin = np.zeros(REPS); tot = np.zeros(REPS)
k = 300
for r in range(REPS):
    torch.cuda.synchronize(); g.sync()
    t0 = time.perf_counter()
    for j in range(K):
        t, ob = scans[k]; k += 1
        ta = time.perf_counter(); g.handle_observation(t, ob); calls[r, j] = time.perf_counter() - ta
    ta = time.perf_counter(); g.sync(); torch.cuda.synchronize(); fin[r] = time.perf_counter() - ta
    tot[r] = time.perf_counter() - t0
us = lambda a: 1e6 * np.median(a, axis=0)
print(f"K={K}: window median {us(tot):.1f} us = {us(tot) / K:.2f} us/update ({K / np.median(tot):.0f} updates/s); closing sync {us(fin):.1f} us; sum of calls {us(calls.sum(axis=1)):.1f} us")
print("per-call host us (median):", " ".join(f"{x:.1f}" for x in us(calls)))
print("flags", g.flags())
