"""GPU suite, round 6 additions (through the C ABI, against the CPU oracle and against the library's own twin launch forms):
  * a scan WITHOUT a matched observation in the middle of the speculation pipeline (round-5 advice: k_mid's m == 0 branch returned in front
    of the next scan's Predict);
  * a filter that can still GROW (capacity as a cap, and the wrappers' default auto-grow) takes the one-launch / speculative form scan
    after scan -- the host learns the n a scan leaves from that scan's k_mid, early (rekf_api.hip, EARLY n) -- with new reflectors turning up
    in the middle of the pipeline;
  * a held scan whose launch fails stays held: the failing call can be repeated, nothing is applied twice or lost.
Tolerances as in test_ekf_gpu.py: association lists identical, |mu - oracle| < 1e-9 (north star: 1e-5 m)."""
import ctypes as C

import numpy as np
import pytest

from reflector_ekf_slam_amd import synth
from reflector_ekf_slam_amd.ekf_slam import RekfError
from tests.helpers import make_oracle, norm_match

pytestmark = pytest.mark.gpu
TIGHT = 1e-9


def _same_match(g, o):
    a, b = norm_match(g.last_match()), norm_match(o.last_match())
    return all(np.array_equal(x, y) for x, y in zip(a, b))


def _counters(g):
    out = (C.c_longlong * 32)()
    assert g._L.rekf_debug_counters(g._h, out) == 0
    return list(out)


def _free_points(sess, k, r0=6.0, clear=1.6):
    """k robot-frame points (at the session's final true pose) whose world positions are at least `clear` metres from every reflector of
    the world and from each other: observations that cannot match anything but themselves."""
    x, y, th = sess.true_pose[-1]
    c, s = np.cos(th), np.sin(th)
    out, world = [], [np.asarray(sess.landmarks, float)]
    ang, r = 0.3, r0
    while len(out) < k:
        rx, ry = r * np.cos(ang), r * np.sin(ang)
        w = np.array([x + c * rx - s * ry, y + s * rx + c * ry])
        if min(float(np.hypot(*(q - w).T).min()) for q in world) >= clear:
            out.append((rx, ry))
            world.append(w[None, :])
        ang += 0.7
        r += 0.35
    return np.asarray(out, np.float32)


def _oracle_from(g, cfg, sess):
    st = g.GetState()
    o = make_oracle(cfg.odom_model, sess.init_time, sess.init_pose, cfg.sigma_v ** 2, cfg.sigma_w ** 2, cfg.sigma_obs ** 2)
    vt = sess.odom[np.nonzero(sess.ev_type == synth.EV_ODOM)[0][-1]]
    o.set_state(st.time, st.mu, st.sigma, vt)
    return o


@pytest.mark.parametrize("K_far", [1, 9], ids=["one_false_positive", "nine_points"])
def test_scan_without_a_match_inside_the_pipeline_full_filter(K_far, monkeypatch):
    """Full filter, scan after scan: every 7th scan consists ONLY of points far from every reflector (a false-positive detection): nothing
    matches, the full filter drops them (capacity flag), k_mid takes its m == 0 branch -- which must still leave the NEXT scan's Predict
    for the speculation pipeline.  The twin without the pipeline (REKF_SPEC=0, REKF_SCAN_LAUNCH=0) must end on the same bits."""
    from reflector_ekf_slam_amd import ReflectorEKFSLAM
    from reflector_ekf_slam_amd import session as S
    cfg = synth.SessionConfig("r6_m0", 110, 16, synth.DIFF, seed=6100 + K_far, speed=1.4, row_spacing=6.0)
    sess = synth.make_session(cfg)
    scans = []
    for k, (t, ob) in enumerate(synth.steady_state_scans(sess, 160)):
        if k % 7 == 3:
            ob = (np.array([[400.0, 300.0]], np.float32) + np.arange(K_far, dtype=np.float32)[:, None] * np.float32(3.0)).astype(np.float32)
        scans.append((t, np.ascontiguousarray(ob, np.float32)))

    def run(pipelined):
        monkeypatch.setenv("REKF_SPEC", "1" if pipelined else "0")
        monkeypatch.setenv("REKF_SCAN_LAUNCH", "1" if pipelined else "0")
        g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
        S.replay(sess, g)
        poses = []
        for k, (t, ob) in enumerate(scans):
            g.handle_observation(t, ob)
            if k % 23 == 22: poses.append(g.pose())
        cnt = _counters(g)
        code = g.sync_code()
        st = g.GetState()
        g.close()
        return st, poses, cnt, code

    a, pa, ca, coa = run(False)
    b, pb, cb, cob = run(True)
    assert ca[20] == 0 and cb[20] > 100, (ca[20], cb[20])
    assert coa == cob == -4                                  # the dropped points: the sticky capacity flag, in both forms
    assert np.array_equal(a.mu, b.mu) and np.array_equal(a.sigma, b.sigma)
    for (ta, ma, sa), (tb, mb, sb) in zip(pa, pb):
        assert ta == tb and np.array_equal(ma, mb) and np.array_equal(sa, sb)


@pytest.mark.parametrize("mode", ["cap_2L", "auto_grow_default"])
def test_growing_filter_takes_the_one_launch_pipeline_and_meets_new_reflectors_in_it(oracle_lib, mode):
    """A filter below its capacity, scan after scan without a read-back: since round 6 it runs the one-launch / speculative form (the host
    learns from each scan's k_mid, early, the n the scan leaves).  From scan 40 on a point that is in nobody's gate shows up in every scan:
    new at its first sight (the pipeline must fall back to the two-launch chain for the scan behind it: downdate, the new rows, then the
    update), matched ever after; a second one at scan 90, and an all-new scan (m == 0 with three appended reflectors) at scan 120.
    Associations identical to the oracle's, |mu - oracle| < 1e-9, and the counters say the pipeline really ran."""
    from reflector_ekf_slam_amd import ReflectorEKFSLAM
    from reflector_ekf_slam_amd import session as S
    cfg = synth.SessionConfig("r6_grow", 120, 18, synth.DIFF, seed=6200, speed=1.4, row_spacing=6.0)
    sess = synth.make_session(cfg)
    if mode == "cap_2L":
        g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=2 * cfg.n_landmarks, auto_grow=False)
    else:
        g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=16)       # auto_grow is the default: 16 -> 32 -> ... -> 256
    S.replay(sess, g)
    o = _oracle_from(g, cfg, sess)
    n0 = o.n
    base = synth.steady_state_scans(sess, 200)
    pts = _free_points(sess, 5)
    extra1, extra2, fresh = pts[0:1], pts[1:2], pts[2:5]
    for k, (t, ob) in enumerate(base):
        ob = np.array(ob, np.float32, copy=True)
        if k == 120:
            ob = fresh.copy()
        else:
            if k >= 40: ob = np.concatenate([ob, extra1])
            if k >= 90: ob = np.concatenate([ob, extra2])
        g.handle_observation(t, ob)
        o.handle_observation(t, ob)
        if k in (40, 41, 42, 90, 91, 120, 121, 199) or k % 37 == 36:
            assert _same_match(g, o), f"association differs at scan {k}"
            assert np.abs(g.mu() - o.mu()).max() < TIGHT, f"mean differs at scan {k}"
    cnt = _counters(g)
    assert o.n == n0 + 2 * 5 and g.n == o.n
    assert cnt[20] > 120 and cnt[24] > 0, (cnt[20], cnt[24])      # speculative scans proved / downdate roles inside k_mid's grid
    st = g.GetState()
    mo, Po = o.state()
    assert np.abs(st.mu - mo).max() < TIGHT and np.abs(st.sigma - Po).max() < 1e-11
    assert g.sync_code() == 0 and g.flags() == 0


def test_growing_filter_pipeline_gives_the_two_launch_chains_bits(monkeypatch):
    """The same growing session with and without the pipeline (REKF_SPEC / REKF_SCAN_LAUNCH off: every scan as front end + k_mid behind the
    previous scan's downdate): bit-identical states at every read-back and at the end -- new reflectors inside the pipeline included."""
    from reflector_ekf_slam_amd import ReflectorEKFSLAM
    from reflector_ekf_slam_amd import session as S
    cfg = synth.SessionConfig("r6_grow_twin", 90, 20, synth.OMNI, seed=6300, speed=1.2, row_spacing=6.0)
    sess = synth.make_session(cfg)
    base = synth.steady_state_scans(sess, 150)
    ex = _free_points(sess, 1)

    def run(pipelined):
        monkeypatch.setenv("REKF_SPEC", "1" if pipelined else "0")
        monkeypatch.setenv("REKF_SCAN_LAUNCH", "1" if pipelined else "0")
        g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=4 * cfg.n_landmarks, auto_grow=False)
        S.replay(sess, g)
        mids = []
        for k, (t, ob) in enumerate(base):
            ob = np.array(ob, np.float32, copy=True)
            if k >= 55: ob = np.concatenate([ob, ex])
            g.handle_observation(t, ob)
            if k % 31 == 30: mids.append(g.GetState())
        cnt = _counters(g)
        assert g.sync_code() == 0
        st = g.GetState()
        g.close()
        return st, mids, cnt

    a, ma, ca = run(False)
    b, mb, cb = run(True)
    assert ca[20] == 0 and cb[20] > 80, (ca[20], cb[20])
    assert a.mu.shape[0] == 3 + 2 * (cfg.n_landmarks + 1)
    assert np.array_equal(a.mu, b.mu) and np.array_equal(a.sigma, b.sigma)
    for x, y in zip(ma, mb):
        assert np.array_equal(x.mu, y.mu) and np.array_equal(x.sigma, y.sigma)


def test_a_held_scan_whose_launch_fails_stays_held(oracle_lib):
    """Scan after scan the newest scan is held on the host and goes out with the next call.  If sending it fails in front of the host's
    bookkeeping (rekf_debug_inject_failure stage 1), the failing call is the one that brought the NEXT scan: it returns the error, neither
    scan has been applied, the held scan is still held -- and repeating the call applies each exactly once (round-5 advice: the old code
    applied the new scan anyway and lost the held one).  A getter that fails the same way leaves the scan held, too."""
    from reflector_ekf_slam_amd import ReflectorEKFSLAM
    from reflector_ekf_slam_amd import session as S
    cfg = synth.SessionConfig("r6_held", 80, 12, synth.DIFF, seed=6400, speed=1.3, row_spacing=6.0)
    sess = synth.make_session(cfg)
    g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
    S.replay(sess, g)
    o = _oracle_from(g, cfg, sess)
    scans = synth.steady_state_scans(sess, 60)
    failed = 0
    for k, (t, ob) in enumerate(scans):
        if k in (9, 10, 25, 40):
            g.inject_failure(1)
            with pytest.raises(RekfError) as err:
                g.handle_observation(t, ob)                  # (sends the held scan k - 1: that fails; scan k is untouched)
            assert err.value.code == -2
            failed += 1
        if k == 33:
            g.inject_failure(1)
            with pytest.raises(RekfError):
                g.pose()                                     # (a getter sends the held scan first: fails, the scan stays held)
            failed += 1
        g.handle_observation(t, ob)
        o.handle_observation(t, ob)
        if k % 10 == 9:
            assert _same_match(g, o), f"association differs at scan {k}"
            assert np.abs(g.mu() - o.mu()).max() < TIGHT, f"mean differs at scan {k}"
    assert failed == 5
    st = g.GetState()
    mo, Po = o.state()
    assert np.abs(st.mu - mo).max() < TIGHT and np.abs(st.sigma - Po).max() < 1e-11
    assert g.sync_code() == 0


def test_map_localisation_at_512_reflectors_through_the_speculative_path(oracle_lib):
    """BASELINE configs[3]'s world (512 reflectors, omni odometry) as the reference's DEPLOYMENT runs it: 400 reflectors come from a
    pre-loaded map (cc:401-425: matched first, by sqrt(e^T S e) < 0.05 against the stored covariances, some of them anisotropic), the
    rest enter the state.  The moving session in the node's call pattern, then 300 scans handed over back to back: since round 6 the
    map branch has a margin proof of its own (RekfDev::map_lip), so these run the speculative one-launch form -- every third one with an
    observation pushed onto the map gate (0.5 m at S = 0.01 I) or the state gate (0.6 m), where the proof must fail and the exact
    re-match decide.  Associations identical to the oracle's, |mu - oracle| < 1e-9."""
    from reflector_ekf_slam_amd import ReflectorEKFSLAM
    from reflector_ekf_slam_amd import session as S
    from tests.helpers import drive_pair
    cfg = synth.C4
    sess = synth.make_session(cfg, max_scans=260)
    rng = np.random.default_rng(6500)
    ids = np.sort(rng.choice(cfg.n_landmarks, size=400, replace=False))
    mxy = (sess.landmarks[ids] + rng.normal(0, 0.01, size=(ids.size, 2))).astype(np.float32)
    mcov = np.tile(np.array([0.01, 0.0, 0.0, 0.01]), (ids.size, 1))
    mcov[::7] = np.array([0.012, 0.003, 0.003, 0.008])                  # symmetric positive definite, anisotropic
    g = ReflectorEKFSLAM(S.options_for(sess))                           # the wrapper's defaults
    o = make_oracle(cfg.odom_model, sess.init_time, sess.init_pose, cfg.sigma_v ** 2, cfg.sigma_w ** 2, cfg.sigma_obs ** 2)
    g.set_map(mxy, mcov); o.set_map(mxy, mcov)
    seen = [0, 0]

    def chk(e, k):
        if k % 25 == 0:
            assert _same_match(g, o), f"association differs at scan {k}"
            m = norm_match(g.last_match())
            seen[0] += m[1].shape[0]; seen[1] += m[0].shape[0]
            assert np.abs(g.mu() - o.mu()).max() < TIGHT

    drive_pair(sess, g, o, chk)
    assert seen[0] > 50 and seen[1] > 10 and g.n == o.n                  # map matches and state matches both occur
    t_park = float(sess.ev_time[-1]) + 0.01                              # (the truncated session ends in motion: park, the scans below are taken standing still)
    g.handle_odometry(t_park, 0.0, 0.0, 0.0); o.handle_odometry(t_park, 0.0, 0.0, 0.0)
    c0 = _counters(g)[20]
    scans = synth.steady_state_scans(sess, 300)
    for k, (t, ob) in enumerate(scans):
        ob = np.array(ob, np.float32, copy=True)
        if k % 3 == 1:
            j = int(rng.integers(0, ob.shape[0]))
            phi = rng.uniform(0, 2 * np.pi)
            r = (0.5 if k % 2 else 0.6) + rng.choice([-4e-3, -1.5e-3, -3e-4, 3e-4, 1.5e-3, 4e-3])
            ob[j] += np.float32(r) * np.array([np.cos(phi), np.sin(phi)], np.float32)
        g.handle_observation(t, ob)
        o.handle_observation(t, ob)
        if k % 29 == 28:
            assert _same_match(g, o), f"association differs at steady scan {k}"
            assert np.abs(g.mu() - o.mu()).max() < TIGHT
    cnt = _counters(g)
    # speculative records proved / with re-matched observations (a third of the gate cases land outside 0.6 m: a new reflector, and the
    # scan behind it takes the two-launch chain; every check above drains the pipeline)
    assert cnt[20] - c0 > 150 and cnt[21] > 10, (c0, cnt[20], cnt[21])
    st = g.GetState()
    mo, Po = o.state()
    assert st.mu.shape == mo.shape and np.abs(st.mu - mo).max() < TIGHT and np.abs(st.sigma - Po).max() < 1e-11
    assert g.sync_code() == 0 and g.flags() == 0


def _readback_session(g, scans, extra_from=None, extra=None, odometry=True):
    """The reference node's call pattern (src/ros_node.cc:514-515, 627-660): odometry messages between the scans, the pose read back after
    every scan.  Returns every pose read and the association lists of every 9th scan."""
    out = []
    t_prev = g.GetLatestTime()
    for k, (t, ob) in enumerate(scans):
        ob = np.array(ob, np.float32, copy=True)
        if extra_from is not None and k >= extra_from:
            ob = np.concatenate([ob, extra])
        if odometry and k % 2 == 0:
            g.handle_odometry(0.5 * (t_prev + t), 0.03, 0.0, 0.004 * ((k % 5) - 2))
        g.handle_observation(t, ob)
        out.append(g.pose())
        if k % 9 == 8:
            out.append(norm_match(g.last_match()))
        t_prev = t
    return out


def _same(a, b):
    if isinstance(a, (tuple, list)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray):
        return np.array_equal(a, b)
    return a == b


@pytest.mark.parametrize("growing", [False, True], ids=["full_filter", "growing_filter"])
def test_grid_match_gives_the_front_kernels_bits_in_the_nodes_call_pattern(oracle_lib, growing):
    """A host-predicted scan (the pose was read back, odometry came in between: the reference node's pattern) needs no front-end launch any
    more: every k_mid workgroup matches it itself from the landmarks in the 3 x 3 match-grid cells around each observation -- exactly the
    reference's first minimum over ALL landmarks, because only a landmark inside 0.6 m can match and those are all in the cells.  Twin:
    the same calls with the grid off (k_front_mb sweeps all L).  Gate cases (observations pushed to 0.6 m +- millimetres), a new
    reflector appearing in the middle (binned where it appears) and, on the growing filter, the oracle's state at the end."""
    from reflector_ekf_slam_amd import ReflectorEKFSLAM
    from reflector_ekf_slam_amd import session as S
    cfg = synth.SessionConfig("r6_grid", 150, 22, synth.DIFF, seed=6600, speed=1.4, row_spacing=6.0)
    sess = synth.make_session(cfg)
    rng = np.random.default_rng(66)
    scans = []
    for k, (t, ob) in enumerate(synth.steady_state_scans(sess, 150)):
        ob = np.array(ob, np.float32, copy=True)
        if k % 4 == 1:
            j = int(rng.integers(0, ob.shape[0]))
            phi = rng.uniform(0, 2 * np.pi)
            r = 0.6 + rng.choice([-3e-3, -1e-3, -2e-4, 2e-4, 1e-3, 3e-3])
            ob[j] += np.float32(r) * np.array([np.cos(phi), np.sin(phi)], np.float32)
        scans.append((t, ob))
    extra = _free_points(sess, 1)

    def run(grid):
        g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=(4 if growing else 1) * cfg.n_landmarks, auto_grow=False)
        g.debug_set_grid(grid)
        S.replay(sess, g)
        seen = _readback_session(g, scans, extra_from=60 if growing else None, extra=extra)
        cnt = _counters(g)
        code = g.sync_code()
        st = g.GetState()
        return g, st, seen, cnt, code

    g0, s0, seen0, c0, code0 = run(False)
    g1, s1, seen1, c1, code1 = run(True)
    assert c0[18] == 0 and c1[18] > 100 and c1[19] == 0, (c0[18], c1[18], c1[19])      # k_mid matched the scans itself, never by the sweep
    assert code0 == code1 and _same(seen0, seen1)
    assert np.array_equal(s0.mu, s1.mu) and np.array_equal(s0.sigma, s1.sigma)
    g0.close(); g1.close()


def test_grid_invalidation_rebuild_and_overflow_fall_back_to_the_same_bits(oracle_lib):
    """The grid is exact only while every landmark stands within the drift bound of where it was binned; a kernel that sees one outside
    marks the grid invalid and tells the host, the scans in between match by the full sweep inside k_mid, the host rebuilds.  With the
    bound lowered to 20 micrometres (rekf_debug_set_grid) that happens every few scans: same bits as the twin without a grid.  And a hash
    table of 8 buckets overflows at once: the library must stop using the grid by itself -- same bits again."""
    from reflector_ekf_slam_amd import ReflectorEKFSLAM
    from reflector_ekf_slam_amd import session as S
    cfg = synth.SessionConfig("r6_grid2", 90, 16, synth.OMNI, seed=6700, speed=1.3, row_spacing=6.0)
    sess = synth.make_session(cfg)
    scans = synth.steady_state_scans(sess, 90)

    def run(on, drift, mask):
        g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
        g.debug_set_grid(on, drift, mask)
        S.replay(sess, g)
        seen = _readback_session(g, scans)
        cnt = _counters(g)
        assert g.sync_code() == 0
        st = g.GetState()
        g.close()
        return st, seen, cnt

    ref, seen_ref, c_ref = run(False, 0.0, -1)
    dr, seen_dr, c_dr = run(True, 2e-5, -1)
    ov, seen_ov, c_ov = run(True, 0.0, 7)
    assert c_ref[18] == 0 and c_ref[17] == 0
    assert c_dr[18] > 60 and c_dr[17] > 20 and c_dr[16] == 1, (c_dr[18], c_dr[17])     # grid-matched scans; rebuilt again and again (every drift note); still in use
    assert c_ov[16] == 0 and c_ov[17] >= 1, (c_ov[16], c_ov[17], c_ov[18])              # the overflowing table was built once or twice and given up
    for st, seen in ((dr, seen_dr), (ov, seen_ov)):
        assert _same(seen_ref, seen)
        assert np.array_equal(ref.mu, st.mu) and np.array_equal(ref.sigma, st.sigma)
