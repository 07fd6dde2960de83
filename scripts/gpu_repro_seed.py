"""Replays ONE seed of scripts/gpu_fuzz_ekf.py's burst mode and prints where |mu - oracle| first leaves the 1e-9 bar; environment:
FUZZ_GRID=0 switches the match grid off, REKF_SPEC / REKF_SCAN_LAUNCH as the library reads them.  python scripts/gpu_repro_seed.py <seed>"""
import json, os, sys
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import synth
from tests.helpers import make_gpu, make_oracle, norm_match
seed = int(sys.argv[1])
every = len(sys.argv) > 2
rng = np.random.default_rng(seed)
model = int(rng.integers(0, 2))
L = int(rng.choice([int(rng.integers(4, 40)), int(rng.integers(40, 140)), int(rng.integers(140, 330))]))
K = int(rng.integers(1, 33))
cfg = synth.SessionConfig(f"fz{seed}", L, K, model, seed=seed, speed=float(rng.uniform(0.5, 2.5)),
                          row_spacing=float(rng.choice([6.0, 9.0, 12.0])), sigma_v=float(rng.uniform(0.02, 0.1)),
                          sigma_w=float(rng.uniform(0.02, 0.1)), sigma_obs=float(rng.uniform(0.03, 0.08)),
                          range_max=float(rng.choice([8.0, 10.0, 14.0])), extra_scans=int(rng.integers(0, 30)))
sess = synth.make_session(cfg, max_scans=int(rng.integers(60, 220)) if L < 140 else int(rng.integers(300, 900)))
lin, ang, obs = cfg.sigma_v ** 2, cfg.sigma_w ** 2, cfg.sigma_obs ** 2
cap = L if (rng.random() < 0.7) else max(4, L // 2)
cap = 2 * L if rng.random() < 0.5 else max(4, L // 2)
g = make_gpu(model, sess.init_time, sess.init_pose, lin, ang, obs, cap)
if cap < L: g.set_auto_grow(True)
if os.environ.get("FUZZ_GRID") == "0": g.debug_set_grid(False)
o = make_oracle(model, sess.init_time, sess.init_pose, lin, ang, obs)
use_map, use_gps = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
assert not use_map
burst_left, scans, first_scan, worst, reported = 0, 0, True, 0.0, False
log = []                                            # the session as fed: (kind, t, payload) -- for off-line witnesses (scripts/ld_witness.py)
for e in range(sess.n_events):
    t = float(sess.ev_time[e])
    if sess.ev_type[e] == synth.EV_ODOM:
        g.handle_odometry(t, *sess.odom[e]); o.handle_odometry(t, *sess.odom[e]); log.append((0, t, np.asarray(sess.odom[e], np.float64))); continue
    if first_scan: first_scan = False; continue
    ob = sess.obs_of(e)
    r = rng.random()
    if r < 0.05: ob = ob[:0]
    elif r < 0.3: ob = ob[: int(rng.integers(1, ob.shape[0] + 1))]
    if rng.random() < 0.03: t -= 0.05
    gps = (sess.true_pose[e] + rng.normal(0, [0.03, 0.03, 0.01])) if (use_gps and rng.random() < 0.5) else None
    g.handle_observation(t, ob, gps); o.handle_observation(t, ob, gps); scans += 1
    log.append((1, t, np.asarray(ob, np.float32).reshape(-1, 2).copy()))
    if not every:
        if burst_left > 0: burst_left -= 1; continue
        burst_left = int(rng.integers(1, 12))
    assert g.sync_code() == 0
    mg, mo = g.mu(), o.mu()
    err = float(np.abs(mg - mo).max())
    if err > 1e-9 and not reported:
        reported = True
        j = int(np.argmax(np.abs(mg - mo)))
        st = g.GetState(); _, Po = o.state()
        dP = np.abs(st.sigma - Po); jj = np.unravel_index(int(np.argmax(dP)), dP.shape)
        print(json.dumps({"first_bad_scan": scans, "n": int(mg.shape[0]), "cap": g.max_landmarks, "err": err, "row": j, "K_this": int(ob.shape[0]),
                          "match": [x.tolist() for x in norm_match(g.last_match())], "dP_max": float(dP.max()), "dP_at": [int(jj[0]), int(jj[1])]}))
    worst = max(worst, err)
if os.environ.get("FUZZ_DUMP"):
    st = g.GetState(); mo, Po = o.state()
    np.savez_compressed(os.environ["FUZZ_DUMP"], kind=np.array([k for k, _, _ in log]), t=np.array([t for _, t, _ in log]),
                        odom=np.array([p if k == 0 else np.zeros(3) for k, _, p in log]),
                        obs_off=np.cumsum([0] + [(p.shape[0] if k == 1 else 0) for k, _, p in log]),
                        obs=np.concatenate([p for k, _, p in log if k == 1] + [np.zeros((0, 2), np.float32)]),
                        gpu_mu=st.mu, gpu_P=st.sigma, oracle_mu=mo, oracle_P=Po, init_pose=np.asarray(sess.init_pose), init_time=sess.init_time,
                        lin=lin, ang=ang, obsv=obs, model=model)
print(json.dumps({"seed": seed, "scans": scans, "worst": worst, "n_final": int(g.mu().shape[0]), "env": {k: v for k, v in os.environ.items() if k.startswith(("REKF_", "FUZZ_"))}}))
