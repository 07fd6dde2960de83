import sys
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import synth
from tests.helpers import make_gpu
seed = 101
rng = np.random.default_rng(seed)
model = int(rng.integers(0, 2)); L = int(rng.integers(10, 60)); K = int(rng.integers(3, 20))
cfg = synth.SessionConfig(f"rnd{seed}", L, K, model, seed=seed, speed=float(rng.uniform(0.5, 2.0)), row_spacing=6.0,
                          sigma_v=float(rng.uniform(0.02, 0.1)), sigma_w=float(rng.uniform(0.02, 0.1)), sigma_obs=float(rng.uniform(0.03, 0.08)))
sess = synth.make_session(cfg, max_scans=120)
g = make_gpu(cfg.odom_model, sess.init_time, sess.init_pose, cfg.sigma_v**2, cfg.sigma_w**2, cfg.sigma_obs**2, cfg.n_landmarks)
use_map = bool(rng.integers(0, 2)); use_gps = bool(rng.integers(0, 2))
print("model", model, "L", L, "K", K, "map", use_map, "gps", use_gps, flush=True)
if use_map:
    ids = rng.choice(L, size=max(2, L // 4), replace=False)
    mxy = (sess.landmarks[ids] + rng.normal(0, 0.01, size=(ids.size, 2))).astype(np.float32)
    mcov = np.tile(np.array([0.01, 0.0, 0.0, 0.01]), (ids.size, 1))
    g.set_map(mxy, mcov)
first = True
for e in range(sess.n_events):
    t = float(sess.ev_time[e])
    if sess.ev_type[e] == synth.EV_ODOM:
        g.handle_odometry(t, *sess.odom[e]); continue
    if first: first = False; continue
    ob = sess.obs_of(e); r = rng.random()
    if r < 0.05: ob = ob[:0]
    elif r < 0.3: ob = ob[: int(rng.integers(1, ob.shape[0] + 1))]
    if rng.random() < 0.03: t -= 0.05
    gps = (sess.true_pose[e] + rng.normal(0, [0.03, 0.03, 0.01])) if (use_gps and rng.random() < 0.5) else None
    print("event", e, "K", ob.shape[0], "gps", gps is not None, "n", end=" ", flush=True)
    g.handle_observation(t, ob, gps)
    g.sync(); m = g.last_match()
    print(g.n, "state", len(m.state_obs_match_ids), "map", len(m.map_obs_match_ids), "new", len(m.new_ids), flush=True)
print("done")
