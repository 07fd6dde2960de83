"""Counters of the match grid after a read-back session (C3, wrapper default): builds, grid-matched scans, sweep fall-backs."""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
cfg = synth.C3
sess = synth.make_session(cfg)
scans = synth.steady_state_scans(sess, 400)
g = ReflectorEKFSLAM(S.options_for(sess))
S.replay(sess, g); g.sync()
def cnt():
    out = (C.c_longlong * 32)(); g._L.rekf_debug_counters(g._h, out); return [int(out[k]) for k in (16, 17, 18, 19)]
print("after map build [on, builds, grid scans, sweeps]:", cnt())
for t, ob in scans[:300]:
    g.handle_observation(t, ob); g.pose()
print("after 300 read-back scans:", cnt())
