"""GPU suite (-m gpu): the HIP path, called through the C ABI (librekf.so via ctypes),
against the CPU oracle on the same seeded inputs and against the committed golden
vectors.  Tolerances: association index lists IDENTICAL; pose / landmarks within
1e-5 m (BASELINE.json north_star) -- in practice we assert far tighter (1e-9).
Nothing here reads /root/reference."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from reflector_ekf_slam_amd import synth
from tests.helpers import drive_pair, make_gpu, make_oracle, norm_match, replay_golden

pytestmark = pytest.mark.gpu

TOL_M = 1e-5          # the north-star tolerance
TIGHT = 1e-9          # what FP64 round-off actually leaves us


def _pair(cfg, sess, cap=None):
    lin, ang, obs = cfg.sigma_v ** 2, cfg.sigma_w ** 2, cfg.sigma_obs ** 2
    g = make_gpu(cfg.odom_model, sess.init_time, sess.init_pose, lin, ang, obs, cap or cfg.n_landmarks)
    o = make_oracle(cfg.odom_model, sess.init_time, sess.init_pose, lin, ang, obs)
    return g, o


def test_extension_is_loaded_and_there_is_no_fallback():
    from reflector_ekf_slam_amd import _lib
    assert os.path.exists(_lib.lib_path("librekf.so"))
    assert _lib.rekf().rekf_abi_version() == _lib.REKF_ABI_VERSION == 7


@pytest.mark.parametrize("case", ["diff_L24_obs8", "omni_L30_obs10", "map_L24_obs8", "gps_L20_obs6", "diff_L128_obs16"])
def test_hip_path_matches_golden_vectors(golden_dir, case):
    g = np.load(os.path.join(golden_dir, case + ".npz"))
    f = make_gpu(int(g["odom_model"]), float(g["init_time"]), g["init_pose"], float(g["lin_cov"]),
                 float(g["ang_cov"]), float(g["obs_cov"]), int(g["n_landmarks"]))
    worst, bad_n, bad_match = replay_golden(g, f)
    assert bad_match == 0 and bad_n == 0
    assert worst < TIGHT
    st = f.GetState()
    assert np.abs(st.mu - g["exp_final_mu"]).max() < TIGHT
    assert np.abs(st.sigma - g["exp_final_sigma"]).max() < 1e-10
    assert f.sync_code() == 0


@pytest.mark.parametrize("cfg", [
    synth.SessionConfig("diff_small", 40, 12, synth.DIFF, seed=21, speed=1.0, row_spacing=6.0),
    synth.SessionConfig("omni_small", 36, 9, synth.OMNI, seed=22, speed=1.0, row_spacing=6.0),
], ids=lambda c: c.name)
def test_trajectory_parity_small(oracle_lib, cfg):
    sess = synth.make_session(cfg, max_scans=250)
    g, o = _pair(cfg, sess)
    worst = [0.0]

    def chk(e, k):
        a, b = norm_match(g.last_match()), norm_match(o.last_match())
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), f"association differs at scan {k}"
        mg, mo = g.mu(), o.mu()
        assert mg.shape == mo.shape
        worst[0] = max(worst[0], float(np.abs(mg - mo).max()))

    drive_pair(sess, g, o, chk)
    assert worst[0] < TIGHT
    st = g.GetState()
    mo, Po = o.state()
    assert np.abs(st.sigma - Po).max() < 1e-12
    # P stays symmetric to round-off without ever being symmetrised (like the reference)
    assert np.abs(st.sigma - st.sigma.T).max() < 1e-13
    assert np.diag(st.sigma).min() >= 0
    assert -math.pi < st.mu[2] <= math.pi


def test_trajectory_parity_c2_full(oracle_lib):
    """BASELINE.json configs[1]: N=128 landmarks, 16 obs/scan, the whole session."""
    cfg = synth.C2
    sess = synth.make_session(cfg)
    g, o = _pair(cfg, sess)
    worst = [0.0]

    def chk(e, k):
        a, b = norm_match(g.last_match()), norm_match(o.last_match())
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), f"association differs at scan {k}"
        if k % 10 == 0:
            worst[0] = max(worst[0], float(np.abs(g.mu() - o.mu()).max()))

    drive_pair(sess, g, o, chk)
    assert g.n == 3 + 2 * cfg.n_landmarks == o.n
    assert worst[0] < TIGHT
    st = g.GetState()
    mo, Po = o.state()
    assert np.abs(st.mu - mo).max() < TIGHT and np.abs(st.sigma - Po).max() < 1e-11
    # landmarks within 1e-5 m of the CPU path (north star) -- trivially, given the above
    assert np.abs(st.mu[3:] - mo[3:]).max() < TOL_M


@pytest.fixture(scope="module")
def c3_built():
    """BASELINE.json configs[2] at FULL size: the map of 1024 landmarks is built on the GPU."""
    from reflector_ekf_slam_amd import session as S
    cfg = synth.C3
    sess = synth.make_session(cfg)
    g = make_gpu(cfg.odom_model, sess.init_time, sess.init_pose, cfg.sigma_v ** 2, cfg.sigma_w ** 2,
                 cfg.sigma_obs ** 2, cfg.n_landmarks)
    # ground-truth identity of every state landmark, from the association stream
    S.replay(sess, g)
    g.sync()
    return cfg, sess, g


def test_c3_full_size_properties(c3_built):
    cfg, sess, g = c3_built
    st = g.GetState()
    n = 3 + 2 * cfg.n_landmarks
    assert st.mu.shape[0] == n and st.sigma.shape == (n, n)
    assert np.isfinite(st.mu).all() and np.isfinite(st.sigma).all()
    # the stored covariance is EXACTLY symmetric: k_downdate2 computes the lower triangle and mirrors it, the predict / augment
    # kernels mirror theirs (partial mirroring -- tiles but not the border strips -- let the antisymmetric part grow without bound)
    assert np.array_equal(st.sigma, st.sigma.T)
    assert np.diag(st.sigma).min() > 0
    assert -math.pi < st.mu[2] <= math.pi
    # size-independent domain property: the map is right -- every estimated landmark sits
    # next to exactly one true reflector and the pose is near the simulated truth
    lm = st.mu[3:].reshape(-1, 2)
    d = np.linalg.norm(lm[:, None] - sess.landmarks[None], axis=-1)
    nearest = d.argmin(1)
    assert len(set(nearest.tolist())) == cfg.n_landmarks          # a bijection: no duplicates
    assert d.min(1).max() < 1.0          # SLAM drift over the 96 m field; reflectors are >= 2 m apart
    assert np.linalg.norm(st.mu[:2] - sess.true_pose[-1][:2]) < 1.0
    # PSD to round-off on a 300x300 principal block (bounded cost).  The reference update is
    # not in Joseph form, so eigenvalues may dip to -eps * lambda_max; they must not go further.
    w = np.linalg.eigvalsh(0.5 * (st.sigma[:300, :300] + st.sigma[:300, :300].T))
    assert w.min() > -1e-12 * w.max()


def test_c3_full_size_steps_match_oracle(c3_built, oracle_lib):
    """From the GPU's own N=1024 state, 12 steady-state updates on both paths."""
    cfg, sess, g = c3_built
    st = g.GetState()
    o = make_oracle(cfg.odom_model, sess.init_time, sess.init_pose, cfg.sigma_v ** 2, cfg.sigma_w ** 2,
                    cfg.sigma_obs ** 2)
    vt = sess.odom[np.nonzero(sess.ev_type == synth.EV_ODOM)[0][-1]]
    o.set_state(st.time, st.mu, st.sigma, vt)
    for t, ob in synth.steady_state_scans(sess, 12):
        g.handle_observation(t, ob)
        o.handle_observation(t, ob)
        a, b = norm_match(g.last_match()), norm_match(o.last_match())
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
        assert a[0].shape[0] == cfg.obs_per_scan and a[2].size == 0
        assert np.abs(g.mu() - o.mu()).max() < TIGHT
    st2 = g.GetState()
    mo, Po = o.state()
    assert np.abs(st2.sigma - Po).max() < 1e-12
    assert g.sync_code() == 0


@pytest.mark.parametrize("L", [1472, 2048])
def test_states_beyond_the_baseline_size_take_the_generic_downdate_loop(oracle_lib, L):
    """n = 2947 / 4099 (T = 46 / 64 tile rows): class-B workgroups of k_downdate2 hold 5 / 11 tiles, i.e. the generic tile
    loop instead of the straight-line code of the BASELINE sizes (<= 4 tiles).  A synthetic state is loaded into both paths
    (landmarks on a grid, a random SPD covariance), then a few updates with 32 observations each."""
    rng = np.random.default_rng(L)
    n = 3 + 2 * L
    side = int(math.ceil(math.sqrt(L)))
    lm = np.array([(2.5 * (k % side), 2.5 * (k // side)) for k in range(L)], dtype=np.float64)
    pose = np.array([2.5 * side / 2 + 0.3, 2.5 * side / 2 - 0.4, 0.3])
    mu = np.concatenate([pose, (lm + rng.normal(0, 0.01, lm.shape)).ravel()])
    A = rng.normal(size=(n, 40))
    P = (A @ A.T) * 1e-5 + np.diag(rng.uniform(1e-4, 4e-4, n))
    P = 0.5 * (P + P.T)
    g = make_gpu(0, 0.0, pose, 0.0025, 0.0064, 0.0025, L)
    o = make_oracle(0, 0.0, pose, 0.0025, 0.0064, 0.0025)
    g.set_state(1.0, mu, P, (0.4, 0.0, 0.1))
    o.set_state(1.0, mu, P, (0.4, 0.0, 0.1))
    c, s_ = math.cos(pose[2]), math.sin(pose[2])
    d = np.linalg.norm(lm - pose[:2], axis=1)
    near = np.argsort(d)[:32]
    for k in range(4):
        rel = lm[near] - pose[:2]
        ob = np.stack([rel[:, 0] * c + rel[:, 1] * s_, -rel[:, 0] * s_ + rel[:, 1] * c], axis=1)
        ob = (ob + rng.normal(0, 0.01, ob.shape)).astype(np.float32)
        t = 1.0 + 0.05 * (k + 1)
        g.handle_observation(t, ob)
        o.handle_observation(t, ob)
        a, b = norm_match(g.last_match()), norm_match(o.last_match())
        assert all(np.array_equal(x, y) for x, y in zip(a, b)) and a[0].shape[0] >= 28
    st = g.GetState()
    mo, Po = o.state()
    assert st.mu.shape[0] == n and g.sync_code() == 0
    assert np.abs(st.mu - mo).max() < TIGHT
    assert np.abs(st.sigma - Po).max() < 1e-11
    assert np.array_equal(st.sigma, st.sigma.T)


# ---------------------------------------------------------------- edge cases / API behaviour
def _simple(cap=8, model=0, pose=(0.0, 0.0, 0.0)):
    return make_gpu(model, 0.0, np.array(pose), 0.0025, 0.0064, 0.0025, cap)


def test_empty_observation_is_predict_only(oracle_lib):
    g = _simple()
    o = make_oracle(0, 0.0, np.zeros(3), 0.0025, 0.0064, 0.0025)
    for f in (g, o):
        f.handle_odometry(0.5, 1.0, 0.0, 0.2)
        f.handle_observation(1.0, np.zeros((0, 2), np.float32))
    assert np.abs(g.mu() - o.mu()).max() < 1e-14
    assert g.GetLatestTime() == 1.0 and g.n == 3
    m = g.last_match()
    assert m.new_ids.size == 0 and m.state_obs_match_ids.size == 0


def test_stale_odometry_and_negative_dt(oracle_lib):
    g = _simple()
    o = make_oracle(0, 0.0, np.zeros(3), 0.0025, 0.0064, 0.0025)
    for f in (g, o):
        f.handle_odometry(1.0, 1.0, 0.0, 0.0)
        f.handle_odometry(0.5, 5.0, 0.0, 0.0)                       # dropped (cc:211-212)
        f.handle_observation(0.9, np.zeros((0, 2), np.float32))     # dt = -0.1 (Q8)
    assert g.GetLatestTime() == 0.9
    assert np.abs(g.mu() - o.mu()).max() < 1e-14


def test_first_scan_all_new_and_float32_means(oracle_lib):
    g = _simple(pose=(1.0, 2.0, 0.3))
    obs = np.array([[1.234567, 2.345678], [2.0, -1.0], [-1.5, 0.7]], np.float32)
    g.handle_observation(0.0, obs)
    st = g.GetState()
    assert st.mu.shape[0] == 9 and list(g.last_match().new_ids) == [0, 1, 2]
    assert all(float(np.float32(v)) == v for v in st.mu[3:])        # Q4
    for a in range(3):
        for b in range(3):
            assert np.allclose(st.sigma[3 + 2 * a: 5 + 2 * a, 3 + 2 * b: 5 + 2 * b], 0.0025 * np.eye(2), atol=1e-18)   # Q7


def test_threshold_edges_and_duplicate_matches():
    g = _simple()
    g.handle_observation(0.0, np.array([[5.0, 0.0]], np.float32))
    g.handle_observation(0.0, np.array([[5.59, 0.0], [5.0, 0.61]], np.float32))
    m = g.last_match()
    assert m.state_obs_match_ids.tolist() == [[0, 0]] and m.new_ids.tolist() == [1]
    g.handle_observation(0.0, np.array([[5.1, 0.0], [4.9, 0.05]], np.float32))
    assert g.last_match().state_obs_match_ids[:, 1].tolist() == [0, 0]      # Q6: both match landmark 0


def test_duplicate_matches_state_parity_incl_chunk_boundary_landmark(oracle_lib):
    """Q6 with numbers: several observations matched to ONE landmark give duplicated H row pairs.  Landmark
    126 owns state rows 255/256, which straddle two 256-row workgroup chunks of k_gather (compact-row table)."""
    g = _simple(cap=160)
    o = make_oracle(0, 0.0, np.zeros(3), 0.0025, 0.0064, 0.0025)
    gx, gy = np.meshgrid(np.arange(15) * 1.5 - 10.0, np.arange(10) * 1.5 - 7.0, indexing="ij")
    pts = np.stack([gx.ravel() + 0.3, gy.ravel() + 0.2], 1).astype(np.float32)          # 150 reflectors, 1.5 m apart
    for f in (g, o):
        for a in range(0, 150, 50):
            f.handle_observation(0.0, pts[a:a + 50])
    assert g.n == o.n == 303
    dup = np.stack([pts[126] + [0.05, 0.0], pts[126] + [-0.04, 0.03], pts[126] + [0.0, -0.05],
                    pts[7] + [0.03, 0.0], pts[7] + [0.0, 0.04], pts[127], pts[60] + [0.01, 0.01]]).astype(np.float32)
    for f in (g, o):
        f.handle_odometry(0.1, 0.2, 0.0, 0.05)
        f.handle_observation(0.2, dup)
    sg, so = norm_match(g.last_match()), norm_match(o.last_match())
    assert sg[0][:, 1].tolist() == [126, 126, 126, 7, 7, 127, 60]
    assert all(np.array_equal(a, b) for a, b in zip(sg, so))
    mg, Pg = g.GetState().mu, g.GetState().sigma
    mo, Po = o.state()
    assert np.abs(mg - mo).max() < 1e-12
    assert np.abs(Pg - Po).max() < 1e-12 * max(1.0, np.abs(Po).max())
    assert g.sync_code() == 0


@pytest.mark.parametrize("L,obs", [(63, 10), (64, 12), (65, 12), (95, 40), (96, 20)],
                         ids=["n129_rem1", "n131_rem3", "n133_rem5_padded", "n193_rem1_m80", "n195_rem3"])
def test_downdate_border_strips_full_covariance(oracle_lib, L, obs):
    """k_downdate treats a border of <= 4 rows past a multiple of 64 (n = 3 + 2L odd: 1 or 3) as strips riding on the
    diagonal tiles instead of padded tiles; 5 rows fall back to padded tiles.  Whole P against the oracle, in the
    single-chunk (m_pad = 64 never here), generic (m_pad < 64) and multi-chunk (m_pad = 80) forms."""
    cfg = synth.SessionConfig(f"strip{L}", L, obs, synth.DIFF, seed=700 + L, speed=2.0, row_spacing=8.0,
                              range_max=16.0 if obs > 20 else 10.0, extra_scans=40)
    sess = synth.make_session(cfg)                    # whole route: the map is complete for the second half
    g, o = _pair(cfg, sess)

    def chk(e, k):
        a, b = norm_match(g.last_match()), norm_match(o.last_match())
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), f"association differs at scan {k}"

    drive_pair(sess, g, o, chk)
    st = g.GetState()
    mo, Po = o.state()
    assert st.mu.shape == mo.shape == (3 + 2 * L,)
    assert np.abs(st.mu - mo).max() < 1e-9
    assert np.abs(st.sigma - Po).max() < 1e-11 * max(1.0, np.abs(Po).max())
    assert np.abs(st.sigma - st.sigma.T).max() < 1e-12
    assert g.sync_code() == 0


def test_marker_ellipses_on_device_match_oracle(oracle_lib):
    """rekf_get_marker_ellipses (src/ros_node.cc:750-765 on the device, 5 doubles per landmark) vs the oracle's
    restatement on the oracle's own state; compared as ellipses (R diag R^T) and component-wise."""
    from tests.helpers import ellipse_matrices
    cfg = synth.SessionConfig("ell", 40, 12, synth.OMNI, seed=33, speed=1.0, row_spacing=6.0)
    sess = synth.make_session(cfg, max_scans=120)
    g, o = _pair(cfg, sess)
    drive_pair(sess, g, o)
    eg, eo = g.marker_ellipses(), o.marker_ellipses()
    L = (o.n - 3) // 2
    assert eg.shape == eo.shape == (L, 5) and L > 8
    assert np.abs(eg[:, :2] - eo[:, :2]).max() < 1e-12                       # landmark means
    Mg, Mo = ellipse_matrices(eg), ellipse_matrices(eo)
    assert np.abs(Mg - Mo).max() <= 1e-9 * np.abs(Mo).max()
    assert np.allclose(eg[:, 3:], eo[:, 3:], rtol=1e-9, atol=0) and np.abs(np.sin(eg[:, 2] - eo[:, 2])).max() < 1e-6
    assert g.marker_ellipses(max_landmarks=L).shape == (L, 5)
    from reflector_ekf_slam_amd import RekfError
    with pytest.raises(RekfError) as e:                                        # caller buffer too small: code, not a crash
        g.marker_ellipses(max_landmarks=3)
    assert e.value.code == -6
    f = _simple()
    assert f.marker_ellipses().shape == (0, 5)                                 # "No reflector detected" (ros_node.cc:741-745)


def test_too_many_observations_is_an_error_code_not_a_crash():
    from reflector_ekf_slam_amd import RekfError
    g = _simple()
    with pytest.raises(RekfError) as e:
        g.handle_observation(0.0, np.zeros((257, 2), np.float32))              # REKF_MAX_OBS = 256 = what the detectors can emit
    assert e.value.code == -3


def test_capacity_overflow_is_reported_and_state_stays_valid():
    g = _simple(cap=2)
    obs = np.array([[3.0, 0.0], [0.0, 3.0], [-3.0, 0.0]], np.float32)
    g.handle_observation(0.0, obs)
    assert g.sync_code() == -4            # REKF_ERR_CAPACITY, sticky-once
    assert g.sync_code() == 0
    assert g.n == 7 and g.last_match().new_ids.tolist() == [0, 1]
    assert np.isfinite(g.GetState().sigma).all()


def test_predict_state_is_non_mutating_and_matches_oracle(oracle_lib):
    g = _simple(pose=(1.0, 1.0, 0.5))
    o = make_oracle(0, 0.0, np.array([1.0, 1.0, 0.5]), 0.0025, 0.0064, 0.0025)
    for f in (g, o):
        f.handle_odometry(0.2, 1.0, 0.0, 0.3)
        f.handle_observation(0.2, np.array([[3.0, 1.0]], np.float32))
    before = g.GetState()
    ps = g.PredictState(0.7)
    after = g.GetState()
    assert np.array_equal(before.mu, after.mu) and np.array_equal(before.sigma, after.sigma)
    mu_p, P_p = o.predict_state(0.7, full=True)
    # the interface's PredictState returns the FULL State (ekf_slam_interface.h:59) with the state's own time (cc:99)
    assert ps.mu.shape == mu_p.shape == (5,) and ps.sigma.shape == (5, 5) and ps.time == before.time == 0.2
    assert np.abs(ps.mu - mu_p).max() < 1e-14 and np.abs(ps.sigma - P_p).max() < 1e-16
    assert np.abs(ps.sigma[:2, 3:] - before.sigma[:2, 3:]).max() > 1e-6      # the pose-landmark cross terms DID move
    pp = g.PredictPose(0.7)                                                    # 96-byte fast path: same pose block
    assert np.array_equal(pp.mu, ps.mu[:3]) and np.array_equal(pp.sigma, ps.sigma[:3, :3])
    t, mu3, s3 = g.pose()
    assert np.array_equal(mu3, after.mu[:3]) and np.array_equal(s3, after.sigma[:3, :3])
    # ... and it equals what the mutating Predict then produces
    g.handle_odometry(0.7, 1.0, 0.0, 0.3)
    st = g.GetState()
    assert np.array_equal(st.mu, ps.mu) and np.array_equal(st.sigma, ps.sigma)


def test_predict_state_full_omni_with_landmarks(oracle_lib):
    cfg = synth.SessionConfig("omni_ps", 20, 8, synth.OMNI, seed=5, speed=1.0, row_spacing=6.0)
    sess = synth.make_session(cfg, max_scans=60)
    g, o = _pair(cfg, sess)
    drive_pair(sess, g, o)
    ps = g.PredictState(g.GetLatestTime() + 0.13)
    mu_p, P_p = o.predict_state(o.time + 0.13, full=True)
    assert ps.mu.shape == mu_p.shape and ps.mu.shape[0] > 11
    assert np.abs(ps.mu - mu_p).max() < 1e-12 and np.abs(ps.sigma - P_p).max() < 1e-13


def test_use_imu_ignores_odometry_like_the_reference():
    """reflector_ekf_slam.cc:213-223: with use_imu the odometry branch is skipped entirely (no vt, no predict, no time)."""
    from reflector_ekf_slam_amd import EKFOptions, ReflectorEKFSLAM
    g = ReflectorEKFSLAM(EKFOptions(use_imu=True, init_pose=(1.0, 2.0, 0.3)), max_landmarks=4, auto_grow=False)
    g.handle_odometry(0.5, 1.0, 0.0, 0.2)
    assert g.GetLatestTime() == 0.0
    t, mu3, s3 = g.pose()
    assert np.array_equal(mu3, [1.0, 2.0, 0.3]) and not s3.any()


def test_sticky_flags_are_visible_without_sync(capfd):
    """Capacity overflow must reach callers that only use the getters (the adapter never calls rekf_sync)."""
    g = _simple(cap=2)
    g.handle_observation(0.1, np.array([[2.0, 0.0], [0.0, 2.0], [-2.0, 0.0]], np.float32))   # 3 new, room for 2
    t, mu3, s3 = g.pose()
    assert g.flags() == 1 and g.flags() == 1                    # REKF_FLAGBIT_CAPACITY, not cleared by reading
    assert "DROPPED" in capfd.readouterr().err
    assert g.sync_code() == -4 and g.flags() == 0


def test_pose_published_by_the_chain_equals_the_state(oracle_lib):
    """rekf_get_pose / rekf_sync read tagged slots in pinned memory that the kernels committing the pose store themselves (the
    tile-(0,0) workgroup of the last k_downdate2 once the state is full; k_front for a caller that reads the pose back at
    odometry rate; a publish kernel otherwise).  Whatever path served it, the pose must be the state's, bit for bit."""
    cfg = synth.SessionConfig("pub", 12, 6, synth.DIFF, seed=5, speed=1.0, row_spacing=5.0)
    sess = synth.make_session(cfg, max_scans=150)
    from reflector_ekf_slam_amd import session as S
    g = make_gpu(cfg.odom_model, sess.init_time, sess.init_pose, cfg.sigma_v ** 2, cfg.sigma_w ** 2, cfg.sigma_obs ** 2, cfg.n_landmarks)

    def check():
        t, mu3, s3 = g.pose()
        st = g.GetState()
        assert np.array_equal(mu3, st.mu[:3]) and np.array_equal(s3, st.sigma[:3, :3]), (mu3, st.mu[:3])

    scans = 0
    for e in range(sess.n_events):
        if sess.ev_type[e] == synth.EV_ODOM:
            g.handle_odometry(sess.ev_time[e], sess.odom[e, 0], sess.odom[e, 1], sess.odom[e, 2])
            if e % 3 == 0:
                check()                    # odometry-rate read-back: the next k_front publishes by itself
        else:
            g.handle_observation(sess.ev_time[e], sess.obs_of(e))
            scans += 1
            if scans % 2 == 0:
                check()
            if scans % 5 == 0:
                g.PredictPose(sess.ev_time[e] + 0.01)     # clobbers the slots with a predicted pose
                check()
    assert g.n == 3 + 2 * cfg.n_landmarks and g.sync_code() == 0      # the map filled up: the folded path was exercised
    check()


def test_set_state_get_state_round_trip_and_two_handles():
    """The device keeps the covariance as its LOWER triangle (column-major, Eigen's order): what comes back is the lower triangle
    of what went in, mirrored.  The witness element sits below the diagonal, so a transposed upload / download would show."""
    rng = np.random.default_rng(0)
    n = 3 + 2 * 20
    A = rng.normal(size=(n, n))
    P = A @ A.T * 1e-3
    P[5, 0] += 1e-9                      # deliberately NOT symmetric: layout/transposition witness (kept: it is below the diagonal)
    P[1, 7] += 1e-9                      # ... and one above the diagonal, which the filter never looks at
    mu = rng.normal(size=n)
    g1, g2 = _simple(cap=32), _simple(cap=24)
    g1.set_state(3.5, mu, P, (0.1, 0.0, 0.2))
    g2.set_state(1.5, 2 * mu, 2 * P)
    s1, s2 = g1.GetState(), g2.GetState()
    Pl = np.tril(P) + np.tril(P, -1).T
    assert Pl[0, 5] == P[5, 0] != P[0, 5] and Pl[1, 7] == P[7, 1] != P[1, 7]
    assert s1.time == 3.5 and np.array_equal(s1.mu, mu) and np.array_equal(s1.sigma, Pl)
    assert np.array_equal(s2.mu, 2 * mu) and np.array_equal(s2.sigma, 2 * Pl)
    t, mu3, s3 = g1.pose()
    assert np.array_equal(mu3, mu[:3]) and np.array_equal(s3, Pl[:3, :3])


def test_c_abi_direct_calls_reject_bad_arguments():
    from reflector_ekf_slam_amd import _lib
    L = _lib.rekf()
    o = _lib.RekfOptions()
    h = C.c_void_p()
    assert L.rekf_create(C.byref(o), 0, 0, C.byref(h)) == -1          # max_landmarks < 1
    assert L.rekf_create(C.byref(o), 4, 0, C.byref(h)) == 0
    n = C.c_int()
    assert L.rekf_get_n(h, C.byref(n)) == 0 and n.value == 3
    buf = (C.c_double * 2)()
    assert L.rekf_get_state(h, None, None, buf, 2, None, 0) == -6     # buffer too small
    assert L.rekf_handle_observation(h, 0.0, None, 3, None) == -1
    L.rekf_destroy(h)


@pytest.mark.parametrize("obs_per_scan,range_max", [(48, 14.0), (64, 16.0), (20, 10.0)], ids=["m96", "m128", "m40"])
def test_large_innovation_blocks(oracle_lib, obs_per_scan, range_max):
    """More than 32 observations per scan: m up to 128 exercises the 8x8-register solve, the multi-chunk
    (non-FAST) downdate pipeline and partially filled k-chunks (m_pad = 48, 96, 128)."""
    cfg = synth.SessionConfig("wide", 160, obs_per_scan, synth.DIFF, seed=41, speed=1.5, row_spacing=12.0,
                              range_max=range_max)
    sess = synth.make_session(cfg, max_scans=140)
    g, o = _pair(cfg, sess)
    worst = [0.0]
    seen = [0]

    def chk(e, k):
        a, b = norm_match(g.last_match()), norm_match(o.last_match())
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), f"association differs at scan {k}"
        seen[0] = max(seen[0], 2 * a[0].shape[0])
        if k % 5 == 0:
            worst[0] = max(worst[0], float(np.abs(g.mu() - o.mu()).max()))

    drive_pair(sess, g, o, chk)
    assert seen[0] >= min(2 * obs_per_scan - 8, 100) or obs_per_scan <= 20
    assert worst[0] < TIGHT
    st = g.GetState()
    mo, Po = o.state()
    assert np.abs(st.sigma - Po).max() < 1e-11 and g.sync_code() == 0


@pytest.mark.parametrize("model,gps", [(synth.DIFF, False), (synth.OMNI, True)], ids=["diff", "omni_gps"])
def test_wide_scans_more_than_64_observations(oracle_lib, model, gps):
    """The reference loops over however many observations a scan holds (cc:397).  Beyond 64 the HIP path matches once and
    runs the joint update as exact block steps of 32 (30 with a pose observation) pairs: same associations, same
    posterior as the oracle's single joint update, including scans that add dozens of landmarks at once, block steps
    without any pair left, and the pose rows riding on the last block step."""
    cfg = synth.SessionConfig("wide_k", 420, 150, model, seed=61, pitch=2.0, jitter=0.3, speed=1.5, row_spacing=10.0,
                              range_max=14.0)
    sess = synth.make_session(cfg, max_scans=90)
    g, o = _pair(cfg, sess)
    kmax, mmax = [0], [0]
    rng = np.random.default_rng(3)
    first = True
    for e in range(sess.n_events):
        t = sess.ev_time[e]
        if sess.ev_type[e] == synth.EV_ODOM:
            g.handle_odometry(t, *sess.odom[e]); o.handle_odometry(t, *sess.odom[e])
            continue
        if first:
            first = False
            continue
        ob = sess.obs_of(e)
        gp = (sess.true_pose[e] + rng.normal(0, 0.02, 3)) if gps else None
        g.handle_observation(t, ob, gp); o.handle_observation(t, ob, gp)
        a, b = norm_match(g.last_match()), norm_match(o.last_match())
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), f"association differs at t={t}"
        kmax[0] = max(kmax[0], ob.shape[0]); mmax[0] = max(mmax[0], a[0].shape[0])
        assert g.n == o.n
        assert np.abs(g.mu() - o.mu()).max() < TIGHT
    assert kmax[0] > 100 and mmax[0] > 70                  # genuinely wide: > 64 observations, > 2 block steps of matches
    st = g.GetState()
    mo, Po = o.state()
    assert np.abs(st.mu - mo).max() < TIGHT and np.abs(st.sigma - Po).max() < 1e-11 and g.sync_code() == 0
    assert np.abs(st.sigma - st.sigma.T).max() < 1e-12


def test_odometry_only_stretches_and_interleaving(oracle_lib):
    """Long runs of HandleOdometryMessage between scans (the single-workgroup predict kernel) interleaved
    with observations (the multi-workgroup front): the pose hand-off between the two paths."""
    cfg = synth.SessionConfig("odo", 30, 8, synth.OMNI, seed=51, speed=1.0, row_spacing=6.0, scan_hz=2.0)
    sess = synth.make_session(cfg, max_scans=40)
    g, o = _pair(cfg, sess)
    drive_pair(sess, g, o)
    assert np.abs(g.mu() - o.mu()).max() < TIGHT
    t, mu3, s3 = g.pose()
    mo, Po = o.state()
    assert np.abs(mu3 - mo[:3]).max() < TIGHT and np.abs(s3 - Po[:3, :3]).max() < 1e-12


@pytest.mark.parametrize("seed", [101, 102, 103, 104, 105, 106])
def test_randomised_sessions_with_map_and_pose_observations(oracle_lib, seed):
    """Randomised small sessions: random model, noise levels, obs counts (ragged, incl. empty scans),
    optional pre-loaded map and pose observations, occasional time jumps backwards (Q8)."""
    rng = np.random.default_rng(seed)
    model = int(rng.integers(0, 2))
    L = int(rng.integers(10, 60))
    K = int(rng.integers(3, 20))
    cfg = synth.SessionConfig(f"rnd{seed}", L, K, model, seed=seed, speed=float(rng.uniform(0.5, 2.0)), row_spacing=6.0,
                              sigma_v=float(rng.uniform(0.02, 0.1)), sigma_w=float(rng.uniform(0.02, 0.1)),
                              sigma_obs=float(rng.uniform(0.03, 0.08)))
    sess = synth.make_session(cfg, max_scans=120)
    g, o = _pair(cfg, sess)
    use_map = bool(rng.integers(0, 2))
    use_gps = bool(rng.integers(0, 2))
    if use_map:
        ids = rng.choice(L, size=max(2, L // 4), replace=False)
        mxy = (sess.landmarks[ids] + rng.normal(0, 0.01, size=(ids.size, 2))).astype(np.float32)
        mcov = np.tile(np.array([0.01, 0.0, 0.0, 0.01]), (ids.size, 1))
        g.set_map(mxy, mcov); o.set_map(mxy, mcov)
    first = True
    worst = 0.0
    for e in range(sess.n_events):
        t = float(sess.ev_time[e])
        if sess.ev_type[e] == synth.EV_ODOM:
            g.handle_odometry(t, *sess.odom[e]); o.handle_odometry(t, *sess.odom[e])
            continue
        if first:
            first = False
            continue
        ob = sess.obs_of(e)
        r = rng.random()
        if r < 0.05:
            ob = ob[:0]                                   # empty cloud: predict only
        elif r < 0.3:
            ob = ob[: int(rng.integers(1, ob.shape[0] + 1))]   # ragged
        if rng.random() < 0.03:
            t -= 0.05                                     # stamped slightly in the past: negative dt
        gps = (sess.true_pose[e] + rng.normal(0, [0.03, 0.03, 0.01])) if (use_gps and rng.random() < 0.5) else None
        g.handle_observation(t, ob, gps); o.handle_observation(t, ob, gps)
        a, b = norm_match(g.last_match()), norm_match(o.last_match())
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), f"association differs at event {e}"
        mg, mo = g.mu(), o.mu()
        assert mg.shape == mo.shape
        worst = max(worst, float(np.abs(mg - mo).max()))
    assert worst < TIGHT
    st = g.GetState()
    mo, Po = o.state()
    assert np.abs(st.sigma - Po).max() < 1e-10 and g.sync_code() == 0


def test_interleaved_sessions_on_one_gpu_do_not_interact(oracle_lib):
    """Fleet serving: several handles (one HIP stream each) fed event by event in round-robin, different sessions and
    models; each must follow its own oracle exactly as if it ran alone."""
    cfgs = [synth.SessionConfig("ms0", 40, 12, synth.DIFF, seed=901, speed=1.0, row_spacing=6.0),
            synth.SessionConfig("ms1", 70, 20, synth.OMNI, seed=902, speed=1.5, row_spacing=6.0),
            synth.SessionConfig("ms2", 33, 8, synth.DIFF, seed=903, speed=0.8, row_spacing=6.0)]
    runs = []
    for cfg in cfgs:
        sess = synth.make_session(cfg, max_scans=90)
        g, o = _pair(cfg, sess)
        runs.append({"sess": sess, "g": g, "o": o, "e": 0, "first": True, "worst": 0.0})
    alive = True
    while alive:
        alive = False
        for r in runs:                                   # one event of every session per round
            sess, g, o = r["sess"], r["g"], r["o"]
            if r["e"] >= sess.n_events:
                continue
            alive = True
            e = r["e"]; r["e"] += 1
            t = float(sess.ev_time[e])
            if sess.ev_type[e] == synth.EV_ODOM:
                g.handle_odometry(t, *sess.odom[e]); o.handle_odometry(t, *sess.odom[e])
                continue
            if r["first"]:
                r["first"] = False
                continue
            ob = sess.obs_of(e)
            g.handle_observation(t, ob); o.handle_observation(t, ob)     # no sync between the sessions: the streams overlap
            if e % 7 == 0:
                a, b = norm_match(g.last_match()), norm_match(o.last_match())
                assert all(np.array_equal(x, y) for x, y in zip(a, b))
                r["worst"] = max(r["worst"], float(np.abs(g.mu() - o.mu()).max()))
    for r in runs:
        mg, mo = r["g"].mu(), r["o"].mu()
        assert mg.shape == mo.shape and np.abs(mg - mo).max() < TIGHT and r["worst"] < TIGHT
        st = r["g"].GetState()
        assert np.abs(st.sigma - r["o"].state()[1]).max() < 1e-10 and r["g"].sync_code() == 0
