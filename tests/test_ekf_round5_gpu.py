"""GPU suite, round 5 additions (through the C ABI, against the CPU oracle and against the library's own other launch forms):
  * the SPECULATIVE ReflectorMatch (scan t + 1's front end inside scan t's launch, proved or repaired by scan t + 1's k_mid) gives the
    exact match's bits -- observations pushed onto the 0.6 m gate included, where the proof must fail and the re-match decide;
  * the state every shipped wrapper defaults to (auto-grow on, capacity 64 growing to the map) at BASELINE configs[2]'s full size, in the
    reference node's call pattern (pose read back after every scan) and scan after scan, against the oracle;
  * several handles driven from several host threads that fire TOGETHER (no launch of the default library contains a workgroup that
    waits for another one: nothing can starve) end bit-identical to a lone session.
Tolerances as in test_ekf_gpu.py: association lists identical, |mu - oracle| < 1e-9 (north star: 1e-5 m)."""
import ctypes as C
import threading
import time

import numpy as np
import pytest

from reflector_ekf_slam_amd import synth
from tests.helpers import make_gpu, make_oracle, norm_match

pytestmark = pytest.mark.gpu
TIGHT = 1e-9


def _same_match(g, o):
    a, b = norm_match(g.last_match()), norm_match(o.last_match())
    return all(np.array_equal(x, y) for x, y in zip(a, b))


def _counters(g):
    out = (C.c_longlong * 32)()
    assert g._L.rekf_debug_counters(g._h, out) == 0
    return list(out)


@pytest.mark.parametrize("L,obs", [(100, 14), (160, 30)], ids=["n203_m28", "n323_m60"])
def test_speculative_match_gives_the_exact_matchs_bits_gate_cases_included(oracle_lib, L, obs, monkeypatch):
    """Scan after scan on a full filter the match of scan t + 1 runs one launch early, against the mean and pose scan t starts from, and
    scan t + 1's k_mid accepts an observation's result only with a margin proof (|d1 - 0.6| and d2 - d1 against the bound of how far the
    update moved pose and reflectors), re-matching the rest exactly.  Every third scan here carries an observation pushed to 0.6 m +- a
    few millimetres from its reflector: the proof cannot hold there.  REKF_SPEC=0 runs the exact front end as a launch of its own;
    both must end on the same bits, and the run must really have speculated AND re-matched."""
    from reflector_ekf_slam_amd import ReflectorEKFSLAM
    from reflector_ekf_slam_amd import session as S
    cfg = synth.SessionConfig("r5_spec", L, obs, synth.DIFF, seed=5600 + L, speed=1.4, row_spacing=6.0)
    sess = synth.make_session(cfg)
    rng = np.random.default_rng(56)
    scans = []
    for k, (t, ob) in enumerate(synth.steady_state_scans(sess, 240)):
        ob = np.array(ob, np.float32, copy=True)
        if k % 3 == 1:
            j = int(rng.integers(0, ob.shape[0]))
            phi = rng.uniform(0, 2 * np.pi)
            r = 0.6 + rng.choice([-4e-3, -1.5e-3, -3e-4, 3e-4, 1.5e-3, 4e-3])
            ob[j] += np.float32(r) * np.array([np.cos(phi), np.sin(phi)], np.float32)
        scans.append((t, ob))

    def run(spec):
        monkeypatch.setenv("REKF_SPEC", "1" if spec else "0")
        g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
        S.replay(sess, g)
        for k, (t, ob) in enumerate(scans):
            g.handle_observation(t, ob)
            if k % 41 == 40: g.pose()                       # a read-back now and then: the pipeline drains and starts again
        cnt = _counters(g)
        code = g.sync_code()
        assert code in (0, -4)                              # (an observation outside the gate is a new reflector the full filter drops: the capacity flag)
        return g.GetState(), norm_match(g.last_match()), cnt[20], cnt[21], code

    ex, m_ex, s0, r0, c0 = run(False)
    sp, m_sp, s1, r1, c1 = run(True)
    assert s0 == 0 and s1 > 150 and r1 > 10, (s0, s1, r1)      # speculative records used / with re-matched observations
    assert c0 == c1 and all(np.array_equal(a, b) for a, b in zip(m_ex, m_sp))
    assert np.array_equal(ex.mu, sp.mu) and np.array_equal(ex.sigma, sp.sigma)
    # (no oracle here: it GROWS where the full filter drops the observations pushed outside the gate; the exact front end's own parity
    # with the oracle is the rest of the suite's business)


def test_the_deployed_default_at_full_size_in_both_call_patterns(oracle_lib):
    """BASELINE.json configs[2] with the wrapper defaults -- auto_grow on, initial capacity 64 (it doubles its way to >= 1024 reflectors) --
    built through the reference's own map build, then 300 steady-state updates with the pose read back after every scan (the reference
    node's pattern, src/ros_node.cc:514-515) and 300 scan after scan: associations identical, |mu - oracle| < 1e-9, sigma to 1e-11."""
    from reflector_ekf_slam_amd import ReflectorEKFSLAM
    from reflector_ekf_slam_amd import session as S
    cfg = synth.C3
    sess = synth.make_session(cfg)
    g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=64)           # auto_grow=True is the default
    S.replay(sess, g)
    st = g.GetState()
    assert st.mu.shape[0] == 3 + 2 * cfg.n_landmarks and g.max_landmarks >= cfg.n_landmarks
    o = make_oracle(cfg.odom_model, sess.init_time, sess.init_pose, cfg.sigma_v ** 2, cfg.sigma_w ** 2, cfg.sigma_obs ** 2)
    vt = sess.odom[np.nonzero(sess.ev_type == synth.EV_ODOM)[0][-1]]
    o.set_state(st.time, st.mu, st.sigma, vt)
    scans = synth.steady_state_scans(sess, 600)
    worst = 0.0
    for k, (t, ob) in enumerate(scans[:300]):                              # the node's pattern
        g.handle_observation(t, ob)
        o.handle_observation(t, ob)
        pose = g.pose()[1]
        worst = max(worst, float(np.abs(pose - o.mu()[:3]).max()))
        if k % 20 == 0:
            assert _same_match(g, o), f"association differs at update {k} (read-back pattern)"
    assert worst < TIGHT
    for k, (t, ob) in enumerate(scans[300:]):                              # scan after scan
        g.handle_observation(t, ob)
        o.handle_observation(t, ob)
        if k % 50 == 49:
            assert _same_match(g, o), f"association differs at update {k} (pipelined)"
            assert np.abs(g.mu() - o.mu()).max() < TIGHT
    cnt = _counters(g)
    assert cnt[20] > 250 and cnt[24] > 0, (cnt[20], cnt[24])               # (round 6) the default filter really ran the speculative one-launch form
    st2 = g.GetState()
    mo, Po = o.state()
    assert st2.mu.shape == mo.shape and np.abs(st2.mu - mo).max() < TIGHT and np.abs(st2.sigma - Po).max() < 1e-11
    assert g.sync_code() == 0 and g.flags() == 0


@pytest.mark.parametrize("full", [True, False], ids=["full_filters_one_launch_per_scan", "growing_filters"])
def test_handles_driven_from_threads_that_fire_together(full):
    """Four handles, four host threads (ctypes releases the GIL inside a call): every round the threads sleep 6 ms -- longer than any
    activity window a heuristic could watch -- meet at a barrier and hand their scans over at the same instant.  By default no launch
    of this library contains a workgroup that waits for another one, so coincident launches can only share the GPU, not starve each
    other: every session ends bit-identical to a lone one, no flag raised."""
    from reflector_ekf_slam_amd import ReflectorEKFSLAM
    from reflector_ekf_slam_amd import session as S
    cfg = synth.SessionConfig("r5_threads", 120, 18, synth.DIFF, seed=5700, speed=1.5, row_spacing=6.0)
    sess = synth.make_session(cfg)
    scans = synth.steady_state_scans(sess, 400)

    def make():
        g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks if full else 8, auto_grow=not full)
        S.replay(sess, g)
        g.sync()
        return g

    lone = make()
    for t, ob in scans:
        lone.handle_observation(t, ob)
    ref = lone.GetState()
    n_thr = 4
    hs = [make() for _ in range(n_thr)]
    bar = threading.Barrier(n_thr)
    errs = []

    def work(g):
        try:
            for k in range(0, len(scans), 2):
                time.sleep(0.006)
                bar.wait()
                for t, ob in scans[k:k + 2]:                # two scans back to back: the second rides the speculation pipeline
                    g.handle_observation(t, ob)
        except Exception as e:                              # pragma: no cover
            errs.append(repr(e))
            bar.abort()

    th = [threading.Thread(target=work, args=(g,)) for g in hs]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for g in hs:
        assert g.sync_code() == 0 and g.flags() == 0
        st = g.GetState()
        assert np.array_equal(st.mu, ref.mu) and np.array_equal(st.sigma, ref.sigma)


def test_every_entry_point_in_the_middle_of_the_pipeline(monkeypatch):
    """Scan after scan on a full filter the newest scan is HELD on the host (it goes out with its successor, whose front end it carries),
    the covariance in memory is one downdate behind and the newest landmark means live in the other buffer.  Every entry point of the C ABI
    must see none of that: called between two scans it returns -- and leaves behind -- exactly what it does on a twin handle that has no
    pipeline at all (REKF_SPEC=0, REKF_SCAN_LAUNCH=0: every scan goes out when it is handed over, as front end + k_mid with the previous
    scan's downdate in front).  Same calls at the same places on both (a call that reads the pose makes the next scan host-predicted --
    the reference's own cos / sin instead of the device's, a round-off level difference DESIGN 3 documents -- so the twin must read
    where the pipelined handle reads).  Wide scans (40 observations: block steps), an empty scan and odometry in between."""
    from reflector_ekf_slam_amd import ReflectorEKFSLAM
    from reflector_ekf_slam_amd import session as S
    cfg = synth.SessionConfig("r5_api", 140, 24, synth.DIFF, seed=5711, speed=1.4, row_spacing=6.0)
    sess = synth.make_session(cfg)
    scans = synth.steady_state_scans(sess, 240)
    wide = {37, 91, 140}                                       # these scans carry 40 observations: block steps through the same kernels

    def run(pipelined):
        monkeypatch.setenv("REKF_SPEC", "1" if pipelined else "0")
        monkeypatch.setenv("REKF_SCAN_LAUNCH", "1" if pipelined else "0")
        g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, auto_grow=False)
        S.replay(sess, g)
        assert g.sync_code() == 0
        seen = []
        t_prev = None
        for k, (t, ob) in enumerate(scans):
            ob = np.array(ob, np.float32, copy=True)
            if k in wide:
                ob = np.concatenate([ob, ob[:16] + np.float32(0.01)])      # (sixteen reflectors seen twice: duplicates in the joint update)
            if k == 61:
                g.handle_observation(t, np.zeros((0, 2), np.float32))      # an empty scan: Predict only (cc:235-236)
            else:
                g.handle_observation(t, ob)
            op = (k // 3) % 16 if k % 3 == 2 else -1          # (two scans in a row run through the pipeline, the third is followed by a call)
            if op == 1: seen.append(("pose", g.pose()))
            elif op == 2: seen.append(("n", g.n))
            elif op == 3: seen.append(("match", norm_match(g.last_match())))
            elif op == 4: st = g.GetState(); seen.append(("state", st.time, st.mu.copy(), st.sigma.copy()))
            elif op == 5: seen.append(("ellipses", g.marker_ellipses().copy()))
            elif op == 6: p = g.PredictState(t + 0.013); seen.append(("predict", p.time, p.mu.copy(), p.sigma.copy()))
            elif op == 7: seen.append(("time", g.GetLatestTime()))
            elif op == 8: seen.append(("flags", g.flags()))
            elif op == 9:
                if t_prev is not None: g.handle_odometry(0.5 * (t + scans[min(k + 1, len(scans) - 1)][0]), 0.02, 0.0, 0.003)
            elif op == 10: seen.append(("mu", g.mu().copy()))
            elif op == 11: seen.append(("pose3", g.PredictPose(t + 0.002).mu.copy()))
            elif op == 12: assert g.sync_code() == 0
            elif op == 13 and k == 119:
                st = g.GetState()                                          # the state out and in again: the pipeline starts over from it
                g.set_state(st.time, st.mu, st.sigma)
            elif op == 14: seen.append(("layout", g.device_layout()[:2]))
            t_prev = t
        code = g.sync_code()
        fin = g.GetState()
        cnt = _counters(g)
        g.close()
        return seen, fin, code, cnt

    def same(a, b):
        if isinstance(a, (tuple, list)):
            return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
        if isinstance(a, np.ndarray):
            return np.array_equal(a, b)
        return a == b

    s0, f0, c0, n0 = run(False)
    s1, f1, c1, n1 = run(True)
    assert n0[20] == 0 and n1[20] > 50, (n0[20], n1[20])     # the twin never speculated; the pipelined run did, scan after scan
    assert c0 == c1
    assert len(s0) == len(s1)
    def worst(a, b):
        if isinstance(a, (tuple, list)):
            return max([worst(x, y) for x, y in zip(a, b)] + [0.0])
        if isinstance(a, np.ndarray):
            return float(np.max(np.abs(np.asarray(a, float) - np.asarray(b, float)))) if a.shape == b.shape and a.size else 0.0
        return abs(float(a) - float(b)) if isinstance(a, (int, float)) and isinstance(b, (int, float)) else 0.0
    for k, (a, b) in enumerate(zip(s0, s1)):
        assert same(a, b), (k, a[0], worst(a[1:], b[1:]))
    assert np.array_equal(f0.mu, f1.mu) and np.array_equal(f0.sigma, f1.sigma) and f0.time == f1.time
