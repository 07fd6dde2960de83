import sys
import numpy as np
sys.path.insert(0, ".")
from reflector_ekf_slam_amd import synth, session as S, ReflectorEKFSLAM
from oracle.binding import OracleEKF
cfg = synth.C2
s = synth.make_session(cfg, max_scans=300)
opt = S.options_for(s)
g = ReflectorEKFSLAM(opt, max_landmarks=cfg.n_landmarks)
o = OracleEKF(cfg.odom_model, s.init_time, s.init_pose, opt.linear_velocity_cov, opt.angular_velocity_cov, opt.observation_cov)
first=True; scans=0; last=0
for e in range(s.n_events):
    if s.ev_type[e] == synth.EV_ODOM:
        g.handle_odometry(s.ev_time[e], *s.odom[e]); o.handle_odometry(s.ev_time[e], *s.odom[e])
    else:
        if first: first=False; continue
        ob = s.obs_of(e)
        g.handle_observation(s.ev_time[e], ob); o.handle_observation(s.ev_time[e], ob)
        scans += 1
        st = g.GetState(); mo, Po = o.state()
        dm = np.abs(st.mu-mo).max(); dP = np.abs(st.sigma-Po).max(); asym = np.abs(st.sigma-st.sigma.T).max()
        if dm > 10*max(last,1e-15) or scans % 25 == 0:
            mm = g.last_match()
            print("scan", scans, "n", st.mu.shape[0], "m", 2*len(mm.state_obs_match_ids), "new", len(mm.new_ids), "dmu %.3e dP %.3e asym %.3e asymO %.3e" % (dm, dP, asym, np.abs(Po-Po.T).max()), "argmax", np.unravel_index(np.abs(st.sigma-Po).argmax(), Po.shape))
            last = dm
