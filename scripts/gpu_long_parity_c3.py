"""Long full-size parity run: C3 (n = 2051, 32 matched observations per scan), N steady-state updates on the GPU and on
the CPU oracle (structured mode) from the same state; association lists compared every scan, mean / covariance at the
end.  GPU box: python scripts/gpu_long_parity_c3.py [updates]"""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
from oracle.binding import OracleEKF
from tests.helpers import norm_match

n_upd = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cfg = synth.C3
sess = synth.make_session(cfg)
g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
S.replay(sess, g); g.sync()
st = g.GetState()
o = OracleEKF(cfg.odom_model, sess.init_time, sess.init_pose, cfg.sigma_v ** 2, cfg.sigma_w ** 2, cfg.sigma_obs ** 2)
vt = sess.odom[np.nonzero(sess.ev_type == 0)[0][-1]]
o.set_state(st.time, st.mu, st.sigma, vt)
scans = synth.steady_state_scans(sess, n_upd)
bad = 0
worst = 0.0
t0 = time.time()
for k, (t, ob) in enumerate(scans):
    g.handle_observation(t, ob); o.handle_observation(t, ob)
    if k % 50 == 0 or k == n_upd - 1:
        a, b = norm_match(g.last_match()), norm_match(o.last_match())
        bad += 0 if all(np.array_equal(x, y) for x, y in zip(a, b)) else 1
        worst = max(worst, float(np.abs(g.mu() - o.mu()).max()))
stg = g.GetState()
_, Po = o.state()
print(json.dumps({"config": cfg.name, "updates": n_upd, "association_mismatches_sampled": bad, "worst_mu_abs_diff": worst,
                  "final_cov_abs_diff": float(np.abs(stg.sigma - Po).max()), "cov_asymmetry_gpu": float(np.abs(stg.sigma - stg.sigma.T).max()),
                  "sync_code": g.sync_code(), "seconds": round(time.time() - t0, 1)}))
