"""Scratch driver for early GPU runs: small-session parity vs the oracle + timing."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
from oracle.binding import OracleEKF

def run(cfg, max_scans, check_every=1):
    s = synth.make_session(cfg, max_scans=max_scans)
    opt = S.options_for(s)
    g = ReflectorEKFSLAM(opt, max_landmarks=cfg.n_landmarks)
    o = OracleEKF(cfg.odom_model, s.init_time, s.init_pose, opt.linear_velocity_cov, opt.angular_velocity_cov, opt.observation_cov)
    worst = [0.0]; bad = [0]
    def chk(e, k):
        if k % check_every: return
        mg = g.last_match(); sp, mp, nw = o.last_match()
        ok = np.array_equal(mg.state_obs_match_ids, sp) and np.array_equal(mg.new_ids, nw) and np.array_equal(mg.map_obs_match_ids, mp)
        if not ok:
            bad[0] += 1
            if bad[0] < 3: print("MATCH MISMATCH at scan", k, mg, sp, nw)
        mu_g = g.mu(); mu_o = o.mu()
        if mu_g.shape != mu_o.shape:
            print("n mismatch", mu_g.shape, mu_o.shape); bad[0] += 1; return
        worst[0] = max(worst[0], float(np.abs(mu_g - mu_o).max()))
    first = True; scans = 0
    for e in range(s.n_events):
        if s.ev_type[e] == synth.EV_ODOM:
            g.handle_odometry(s.ev_time[e], *s.odom[e]); o.handle_odometry(s.ev_time[e], *s.odom[e])
        else:
            if first: first = False; continue
            ob = s.obs_of(e)
            g.handle_observation(s.ev_time[e], ob); o.handle_observation(s.ev_time[e], ob)
            scans += 1; chk(e, scans)
    st = g.GetState(); mo, Po = o.state()
    print(cfg.name, "scans", scans, "n", st.mu.shape[0], "bad", bad[0], "max|mu diff|", worst[0],
          "final |P diff|", float(np.abs(st.sigma - Po).max()), "rc", g.sync_code())
    return g, s

if __name__ == "__main__":
    cfg = synth.SessionConfig("tiny", 24, 8, synth.DIFF, seed=7, speed=1.0, row_spacing=6.0)
    run(cfg, 200)
    run(synth.C2, 300, check_every=5)
