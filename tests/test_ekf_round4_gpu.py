"""GPU suite, round 4 additions (through the C ABI, against the CPU oracle):
  * a failing HandleObservationMessage leaves the handle as it found it (fault injection, include/rekf_debug.h): the scan handed
    over again gives the oracle's state -- the host's counters (scan parity, front-end target, growth bound, publisher tags),
    the pose mirror and the time only move once nothing but kernel launches is left;
  * every wrapper's default is the reference's behaviour: max_landmarks is the INITIAL capacity, no reflector is ever dropped
    (reflector_ekf_slam.cc:316-363 resizes on every augment);
  * rekf_device_layout / rekf_sync hand out a device state that includes the predicts the host applied to its pose mirror only.
Tolerances as in test_ekf_gpu.py: association lists identical, |mu - oracle| < 1e-9 (north star: 1e-5 m)."""
import ctypes as C

import numpy as np
import pytest

from reflector_ekf_slam_amd import synth
from reflector_ekf_slam_amd.ekf_slam import RekfError
from tests.helpers import make_gpu, make_oracle, norm_match

pytestmark = pytest.mark.gpu
TIGHT = 1e-9


def _same_match(g, o):
    a, b = norm_match(g.last_match()), norm_match(o.last_match())
    return all(np.array_equal(x, y) for x, y in zip(a, b))


def _session(L=60, obs=10, seed=4401):
    cfg = synth.SessionConfig(f"r4_L{L}", L, obs, synth.DIFF, seed=seed, speed=1.2, row_spacing=6.0)
    return cfg, synth.make_session(cfg)


@pytest.mark.parametrize("readback", [False, True], ids=["pipelined", "pose_read_back"])
def test_a_failing_scan_leaves_the_handle_as_it_found_it(oracle_lib, readback):
    """Stages 1 (staging a wide scan) and 2 (enqueueing the held-back downdate) fail BEFORE anything of the handle has moved: the
    call returns REKF_ERR_HIP, the same scan handed over again is applied once, and the session ends on the oracle's state.  Stage
    3 fails at the launch check BEHIND the chain: the scan has been applied, the handle resynchronises with the device."""
    cfg, sess = _session()
    lin, ang, ob2 = cfg.sigma_v ** 2, cfg.sigma_w ** 2, cfg.sigma_obs ** 2
    g = make_gpu(cfg.odom_model, sess.init_time, sess.init_pose, lin, ang, ob2, cfg.n_landmarks)
    o = make_oracle(cfg.odom_model, sess.init_time, sess.init_pose, lin, ang, ob2)
    first, scans, injected = True, 0, {1: 0, 2: 0, 3: 0}
    for e in range(sess.n_events):
        t = sess.ev_time[e]
        if sess.ev_type[e] == synth.EV_ODOM:
            g.handle_odometry(t, *sess.odom[e]); o.handle_odometry(t, *sess.odom[e])
            continue
        if first:
            first = False
            continue
        ob = sess.obs_of(e)
        scans += 1
        stage = {5: 1, 9: 2, 14: 3, 40: 2, 41: 1, 57: 3, 58: 2}.get(scans, 0)
        if stage:
            if stage == 2:
                g.pose()                                 # (rounds 3-4: only behind a read-back did the held-back downdate go out alone; now the injection itself sees to it)
            g.inject_failure(stage)
            with pytest.raises(RekfError) as err:
                g.handle_observation(t, ob)
            assert err.value.code == -2
            injected[stage] += 1
            if stage != 3:
                g.handle_observation(t, ob)              # not applied: hand it over again
        else:
            g.handle_observation(t, ob)
        o.handle_observation(t, ob)
        if readback:
            _, pg, _ = g.pose()
            assert np.abs(pg - o.mu()[:3]).max() < TIGHT, f"pose differs at scan {scans}"
        if scans % 5 == 0 or stage:
            assert _same_match(g, o), f"association differs at scan {scans}"
            assert np.abs(g.mu() - o.mu()).max() < TIGHT, f"mean differs at scan {scans}"
    assert all(v > 0 for v in injected.values()) and g.n == o.n
    st = g.GetState()
    mo, Po = o.state()
    assert np.abs(st.mu - mo).max() < TIGHT and np.abs(st.sigma - Po).max() < 1e-11
    assert g.sync_code() == 0


def test_inject_stage_2_sends_the_held_back_downdate_out_alone_and_fails_it(oracle_lib):
    """Since round 5 the held-back downdate of a default handle always travels with the next scan's launch (k_dd_front, or the scan's
    one launch) -- also behind a pose read-back.  rekf_debug_inject_failure(2) therefore makes the next scan send it out ALONE, as an
    exclusive handle does behind a read-back, and fails that launch before anything of the handle has moved: the call returns
    REKF_ERR_HIP, the same scan handed over again is applied once."""
    cfg, sess = _session(L=30, obs=8, seed=4402)
    lin, ang, ob2 = cfg.sigma_v ** 2, cfg.sigma_w ** 2, cfg.sigma_obs ** 2
    g = make_gpu(cfg.odom_model, sess.init_time, sess.init_pose, lin, ang, ob2, cfg.n_landmarks)
    o = make_oracle(cfg.odom_model, sess.init_time, sess.init_pose, lin, ang, ob2)
    scans = [(sess.ev_time[e], sess.obs_of(e)) for e in range(sess.n_events) if sess.ev_type[e] != synth.EV_ODOM][1:8]
    raised = 0
    for k, sc in enumerate(scans):
        if k in (1, 4):
            if k == 4: g.pose()                          # (... and behind a read-back)
            g.inject_failure(2)
        # (round 6: scan after scan the newest scan is HELD on the host and sent by the next call -- its failure is then that call's, whose
        # own scan has not been touched: whichever call reports the error is simply repeated)
        try:
            g.handle_observation(*sc)
        except RekfError as e:
            assert e.code == -2
            raised += 1
            g.handle_observation(*sc)
        o.handle_observation(*sc)
    assert raised == 2
    assert g.sync_code() == 0 and g.n == o.n and np.abs(g.mu() - o.mu()).max() < TIGHT


def test_default_wrapper_capacity_grows_like_the_reference(oracle_lib):
    """ReflectorEKFSLAM() with default arguments: max_landmarks is where the buffers START (every wrapper's default, rekf.h)."""
    from reflector_ekf_slam_amd import EKFOptions, ReflectorEKFSLAM
    cfg, sess = _session(L=40, obs=9, seed=4403)
    lin, ang, ob2 = cfg.sigma_v ** 2, cfg.sigma_w ** 2, cfg.sigma_obs ** 2
    opt = EKFOptions(use_imu=False, init_time=float(sess.init_time), init_pose=tuple(float(v) for v in sess.init_pose),
                     odom_model=int(cfg.odom_model), linear_velocity_cov=lin, angular_velocity_cov=ang, observation_cov=ob2)
    g = ReflectorEKFSLAM(opt, max_landmarks=8)           # far too small on purpose
    o = make_oracle(cfg.odom_model, sess.init_time, sess.init_pose, lin, ang, ob2)
    from tests.helpers import drive_pair
    drive_pair(sess, g, o)
    assert g.n == o.n == 3 + 2 * 40 and g.flags() == 0 and g.max_landmarks >= 40
    assert np.abs(g.mu() - o.mu()).max() < TIGHT


def test_device_layout_includes_the_predicts_the_host_applied_to_its_mirror(oracle_lib):
    """Odometry messages launch nothing (host pose mirror).  rekf_device_layout / rekf_sync must still hand raw-pointer users the
    state the getters return: P's columns 0, 1, the pose block and mu[0..2] with those predicts applied."""
    cfg, sess = _session(L=20, obs=6, seed=4404)
    lin, ang, ob2 = cfg.sigma_v ** 2, cfg.sigma_w ** 2, cfg.sigma_obs ** 2
    g = make_gpu(cfg.odom_model, sess.init_time, sess.init_pose, lin, ang, ob2, cfg.n_landmarks)
    o = make_oracle(cfg.odom_model, sess.init_time, sess.init_pose, lin, ang, ob2)
    from tests.helpers import drive_pair
    drive_pair(sess, g, o, max_events=sess.n_events // 2)
    t = g.GetLatestTime()
    for k in range(3):                                   # predicts on the mirror only
        g.handle_odometry(t + 0.02 * (k + 1), 0.4, 0.0, 0.2); o.handle_odometry(t + 0.02 * (k + 1), 0.4, 0.0, 0.2)
    ld, nmax, Pp, mup = C.c_int(), C.c_int(), C.c_void_p(), C.c_void_p()
    assert g._L.rekf_device_layout(g._h, C.byref(ld), C.byref(nmax), C.byref(Pp), C.byref(mup)) == 0
    assert g.sync_code() == 0
    n = g.n
    mu_raw = np.zeros(n)
    P_raw = np.zeros((ld.value, ld.value))
    hip = C.CDLL("libamdhip64.so")                       # (already mapped: librekf.so links it)
    assert hip.hipMemcpy(mu_raw.ctypes.data_as(C.c_void_p), mup, C.c_size_t(8 * n), 2) == 0
    assert hip.hipMemcpy(P_raw.ctypes.data_as(C.c_void_p), Pp, C.c_size_t(8 * ld.value * ld.value), 2) == 0
    Pcol = P_raw.reshape(ld.value, ld.value).T            # column-major on the device
    mo, Po = o.state()
    assert np.abs(mu_raw - mo).max() < TIGHT
    low = np.tril_indices(n)
    assert np.abs(Pcol[:n, :n][low] - Po[low]).max() < 1e-11
    st = g.GetState()
    assert np.array_equal(st.mu, mu_raw)


@pytest.mark.parametrize("readback", [True, False], ids=["pose_read_back", "pipelined"])
def test_several_sessions_on_one_gpu_end_bit_identical_to_a_lone_one(readback):
    """Two of the round-4 launches contain workgroups that WAIT for other workgroups of the same launch (k_mid's mid workgroups for the
    scan's front end / for the previous scan's augmentation).  Alone on the GPU that is deadlock-free; with several sessions enqueueing at
    once the waiting workgroups of one can hold the CUs another one's working workgroups need (scripts/gpu_stress_sessions.py found it:
    six sessions, timed-out waits, garbage).  The library therefore uses the in-launch hand-overs only while no other handle of the
    process is at work (rekf_api.hip: WHO ELSE IS AT WORK ON THE GPU): six handles driven round-robin -- growing filters, with and without
    read-backs -- must each end bit-identical to a handle that ran the same session alone, without a sticky flag."""
    from reflector_ekf_slam_amd import ReflectorEKFSLAM
    from reflector_ekf_slam_amd import session as S
    cfg = synth.SessionConfig("r4_sessions", 160, 20, synth.DIFF, seed=4410, speed=1.5, row_spacing=6.0)
    sess = synth.make_session(cfg)

    def run(handles):
        first = True
        for e in range(sess.n_events):
            t = float(sess.ev_time[e])
            if sess.ev_type[e] == synth.EV_ODOM:
                for g in handles:
                    g.handle_odometry(t, *sess.odom[e])
                continue
            if first:
                first = False
                continue
            ob = sess.obs_of(e)
            for g in handles:
                g.handle_observation(t, ob)
            if readback:
                for g in handles:
                    g.pose()
        for g in handles:
            assert g.sync_code() == 0 and g.flags() == 0
        return [g.GetState() for g in handles]

    lone = run([ReflectorEKFSLAM(S.options_for(sess), max_landmarks=8)])[0]          # (default wrapper: grows 8 -> 16 -> ... on the way)
    many = run([ReflectorEKFSLAM(S.options_for(sess), max_landmarks=8) for _ in range(6)])
    assert lone.mu.shape[0] == 3 + 2 * 160
    for st in many:
        assert np.array_equal(st.mu, lone.mu) and np.array_equal(st.sigma, lone.sigma)


@pytest.mark.parametrize("exclusive", [False, True], ids=["front_end_as_a_launch", "exclusive_front_end_in_the_grid"])
@pytest.mark.parametrize("L,obs", [(100, 14), (160, 30)], ids=["n203_m28", "n323_m60"])
def test_one_launch_per_scan_gives_the_two_launch_chains_bits(oracle_lib, L, obs, exclusive, monkeypatch):
    """Round 5: on a filter that cannot grow the held-back downdate of scan t runs as a ROLE of scan t+1's k_mid launch (from the stored
    P into the other P buffer, tiles from a queue) while that launch's mid role corrects what it gathers by the pending panels, in the
    downdate's own arithmetic.  REKF_SCAN_LAUNCH=0 sends the round-4 chain instead (k_dd_front in place, then k_mid on the downdated
    P).  Same bits -- scan after scan, with pose read-backs and full-state read-backs in between -- and the oracle's state; and the
    in-launch form must really have been used (the tile queue counted)."""
    import ctypes as C
    from reflector_ekf_slam_amd import ReflectorEKFSLAM
    from reflector_ekf_slam_amd import session as S
    cfg = synth.SessionConfig("r5_scan_launch", L, obs, synth.DIFF, seed=5500 + L, speed=1.4, row_spacing=6.0)
    sess = synth.make_session(cfg)
    lin, ang, ob2 = cfg.sigma_v ** 2, cfg.sigma_w ** 2, cfg.sigma_obs ** 2
    scans = synth.steady_state_scans(sess, 200)

    def run(scan_launch):
        monkeypatch.setenv("REKF_SCAN_LAUNCH", "1" if scan_launch else "0")
        g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
        g.set_exclusive(exclusive)
        S.replay(sess, g)                                   # the map; the filter is full afterwards
        mids = []
        for k, (t, ob) in enumerate(scans):
            g.handle_observation(t, ob)
            if k % 37 == 36: g.pose()                       # a pose read-back: the next scan is host-predicted, the downdate stays pending
            if k % 61 == 60: mids.append(g.GetState())      # a full read-back: the pending downdate is applied in place first
        out = (C.c_longlong * 32)()
        assert g._L.rekf_debug_counters(g._h, out) == 0
        assert g.sync_code() == 0 and g.flags() == 0
        return g.GetState(), mids, int(out[24])

    two, mids_two, q_two = run(False)
    one, mids_one, q_one = run(True)
    assert q_two == 0 and q_one > 0
    assert one.mu.shape[0] == 3 + 2 * L
    assert np.array_equal(one.mu, two.mu) and np.array_equal(one.sigma, two.sigma)
    for a, b in zip(mids_one, mids_two):
        assert np.array_equal(a.mu, b.mu) and np.array_equal(a.sigma, b.sigma)
    o = make_oracle(cfg.odom_model, sess.init_time, sess.init_pose, lin, ang, ob2)
    S.replay(sess, o)
    for t, ob in scans:
        o.handle_observation(t, ob)
    mo, Po = o.state()
    assert one.mu.shape == mo.shape and np.abs(one.mu - mo).max() < TIGHT and np.abs(one.sigma - Po).max() < 1e-11
