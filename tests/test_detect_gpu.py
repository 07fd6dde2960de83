"""GPU suite for the 2D/3D reflector detectors: the HIP path through the C ABI (librdet.so)
against the CPU oracle on the same scans.  2D: centres and de-skewed returns are BIT-IDENTICAL to the oracle's
(since round 3 the device evaluates float32 sin / cos with glibc's own algorithm, csrc/glibc_sincosf.h; the host must have
FMA, as every x86-64-v3 CPU does, for its libm to take the same path).  3D: identical clusters in identical order; TOL (the
north-star 1e-5 m) is what the 3D centres and the pipeline tests are held to."""
import math

import numpy as np
import pytest

from tests.detect_cases import S2B, beams_for_width, odom_stream, plate_scan, world_scan

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _pair(s2b=(0.0, 0.0, 0.0), **kw):
    from oracle.binding import OracleDetect2D
    from reflector_ekf_slam_amd.detect import LaserReflectorDetect, ReflectorDetectOptions
    g = LaserReflectorDetect(ReflectorDetectOptions(**kw), max_beams=8192, sensor_to_base_link=s2b)
    o = OracleDetect2D(sensor_to_base_link=s2b, **kw)
    return g, o


def _scan_msg(sc):
    from reflector_ekf_slam_amd.detect import LaserScan
    return LaserScan(sc.stamp, sc.angle_min, sc.angle_max, sc.angle_increment, sc.scan_time, sc.range_min,
                     sc.range_max, sc.ranges, sc.intensities)


def _feed_odom(g, o, stream):
    from reflector_ekf_slam_amd import OdometryData
    for (t, px, py, qz, qw, vx, vy, wz) in stream:
        g.HandleOdometryData(OdometryData(t, (vx, vy, 0.0), (0.0, 0.0, wz), (px, py, 0.0), (qw, 0.0, 0.0, qz)))
        o.handle_odometry(t, px, py, qz, qw, vx, vy, wz)


def _compare(g, o, sc, check_returns=True):
    obs = g.HandleLaserScan(_scan_msg(sc))
    t, c = o.handle_scan(sc)
    assert obs.time_ == t
    assert obs.cloud_.shape == c.shape, (obs.cloud_.shape, c.shape)
    assert np.array_equal(obs.cloud_, c), float(np.abs(obs.cloud_ - c).max())          # bit for bit
    if check_returns:
        rg, ro = g.GetRangeData().returns, o.returns()
        assert rg.shape == ro.shape and np.array_equal(rg, ro)
    return obs


def test_world_scans_static_and_moving(oracle_lib):
    for seed, pose in ((1, (16.0, 17.7, 0.6)), (2, (8.0, 30.0, -2.0)), (3, (25.0, 9.0, 3.0))):
        sc, _ = world_scan(seed=seed, pose=pose)
        g, o = _pair(S2B)
        obs = _compare(g, o, sc)
        assert obs.cloud_.shape[0] >= 15
        g2, o2 = _pair(S2B)
        _feed_odom(g2, o2, odom_stream(sc.stamp - 0.3, sc.stamp + 0.05))
        _compare(g2, o2, sc)
        # a second scan through the same handles (angle table cached, odometry trimmed)
        sc.stamp += 0.1
        _feed_odom(g2, o2, odom_stream(sc.stamp - 0.02, sc.stamp + 0.05, v=0.8, w=-0.2))
        _compare(g2, o2, sc)


def test_state_machine_cases_match(oracle_lib):
    n, rng_ = 720, 5.0
    nb = beams_for_width(0.18, rng_, n)
    cases = []
    cases.append(plate_scan(n, [(100, nb, rng_, 200.0), (300, 3 * nb, rng_, 200.0)]))         # width gate
    sc = plate_scan(n, [(100, nb, rng_, 200.0)]); sc.intensities[102] = 50.0; cases.append(sc)   # bridged gap
    sc = plate_scan(n, [(100, nb, rng_, 200.0)]); sc.intensities[102] = 50.0; sc.ranges[102] = 9.0
    sc.ranges[103] = 5.4; cases.append(sc)                                                       # gap not bridged
    sc = plate_scan(n, [(100, nb, rng_, 200.0)]); sc.intensities[102] = 50.0; sc.ranges[102] = np.inf
    cases.append(sc)                                                                             # inf gap beam
    cases.append(plate_scan(n, [(n - 2, 2, rng_, 200.0), (0, 3, rng_, 200.0), (200, nb, rng_, 200.0)]))   # seam union
    cases.append(plate_scan(n, [(0, 2, rng_, 200.0), (200, nb, rng_, 200.0)]))                   # circle clause
    cases.append(plate_scan(n, [(n - nb, nb, rng_, 200.0)]))                                     # open last run only
    cases.append(plate_scan(n, [(50, nb, rng_, 200.0), (n - nb, nb, rng_, 200.0)]))              # closed + open tail
    cases.append(plate_scan(360, []))                                                            # nothing bright
    sc = plate_scan(n, [(100, nb, rng_, 200.0)]); sc.ranges[:50] = 100.0; cases.append(sc)       # beams outside msg range
    for i, sc in enumerate(cases):
        g, o = _pair()
        _compare(g, o, sc)


def test_ragged_sizes_and_many_reflectors(oracle_lib):
    rng = np.random.default_rng(5)
    for n in (1, 2, 63, 1025, 3601, 8192):
        rng_ = 4.0
        nb = max(beams_for_width(0.18, rng_, n), 1)
        plates = [(int(s), nb, rng_, 200.0) for s in range(5, max(n - nb - 5, 6), max(4 * nb, 8))][:200]
        sc = plate_scan(n, plates if n > 64 else [])
        sc.ranges += rng.normal(0, 0.002, size=n).astype(np.float32)
        g, o = _pair()
        _compare(g, o, sc)


def test_invalid_stretches_and_bright_beams_outside_the_message_range(oracle_lib):
    """What the one-launch kernel's per-beam workgroups have to find on their own: the first and the last valid beam of the
    scan (probed 64 beams at a time from either end), the "a point exists" guard in front of the first valid beam, bright
    beams beyond the message's own range limits (they take point_cloud.back()), an empty point cloud."""
    n, rng_ = 3000, 5.0
    nb = beams_for_width(0.18, rng_, n)
    cases = []
    sc = plate_scan(n, [(100, nb, rng_, 200.0), (900, nb, rng_, 200.0), (2000, nb, rng_, 200.0)])
    sc.ranges[:700] = 100.0                         # first valid beam in the third 256-beam group; the plate at 100 is bright but has no point
    cases.append(sc)
    sc = plate_scan(n, [(900, nb, rng_, 200.0), (2000, nb, rng_, 200.0)])
    sc.ranges[2300:] = np.inf                       # last valid beam ~700 from the end
    cases.append(sc)
    sc = plate_scan(n, [(900, nb, rng_, 200.0), (2000, nb, rng_, 200.0)])
    sc.ranges[905] = 45.0; sc.ranges[2003] = 45.0   # bright (intensity 200, inside the detector's range gate) but beyond range_max = 30
    cases.append(sc)
    sc = plate_scan(n, [(900, nb, rng_, 200.0)])
    sc.ranges[:] = 100.0                            # nothing valid at all
    cases.append(sc)
    sc = plate_scan(n, [(900, nb, rng_, 200.0)])
    sc.ranges[:] = np.inf; sc.ranges[1500] = 7.0    # a single valid beam
    cases.append(sc)
    for sc in cases:
        for with_odom in (False, True):
            g, o = _pair(S2B, range_max=60.0)
            if with_odom:
                _feed_odom(g, o, odom_stream(sc.stamp - 0.3, sc.stamp + 0.05))
            _compare(g, o, sc)


def test_detectors_without_the_pcie_bar_path(oracle_lib):
    """RDET_NO_BAR=1 forces the fallback of platforms whose BAR does not cover device memory (pinned staging + copy instead
    of the host writing the scan / cloud straight into fine-grained device memory): same results, in a fresh process."""
    import os, subprocess, sys
    code = (
        "import sys; sys.path.insert(0, '.')\n"
        "import numpy as np\n"
        "from tests.test_detect_gpu import _pair, _compare, _feed_odom\n"
        "from tests.detect_cases import S2B, world_scan, odom_stream\n"
        "from reflector_ekf_slam_amd import synth\n"
        "from reflector_ekf_slam_amd.detect import PointCloudOptions, PointCloudReflectorDetect\n"
        "from oracle.binding import oracle_detect3d\n"
        "sc, _ = world_scan(seed=4, pose=(12.0, 20.0, 1.0))\n"
        "g, o = _pair(S2B)\n"
        "_feed_odom(g, o, odom_stream(sc.stamp - 0.3, sc.stamp + 0.05))\n"
        "for k in range(3):\n"
        "    obs = _compare(g, o, sc); sc.stamp += 0.1\n"
        "assert obs.cloud_.shape[0] >= 10\n"
        "rng = np.random.Generator(np.random.PCG64(3))\n"
        "lms = synth.make_world(synth.C4, rng)\n"
        "cloud = synth.make_point_cloud(lms, (float(lms[:, 0].mean()), float(lms[:, 1].mean()), 0.3), rng, rings=16, n_az=900)\n"
        "d3 = PointCloudReflectorDetect(PointCloudOptions(), max_points=cloud.shape[0])\n"
        "for k in range(2):\n"
        "    ob3 = d3.HandlePointCloud(1.0 + k, cloud)\n"
        "co, _, _ = oracle_detect3d(cloud)\n"
        "assert ob3.cloud_.shape == co.shape and np.array_equal(ob3.cloud_, co)\n"
        "print('fallback ok', obs.cloud_.shape[0], co.shape[0])\n")
    env = dict(os.environ, RDET_NO_BAR="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0 and "fallback ok" in out.stdout, out.stderr[-3000:]


def test_bad_scan_and_capacity_are_error_codes(oracle_lib):
    from reflector_ekf_slam_amd.detect import LaserReflectorDetect, RdetError, ReflectorDetectOptions
    g = LaserReflectorDetect(ReflectorDetectOptions(), max_beams=256)
    sc = plate_scan(360, [])
    with pytest.raises(RdetError) as e:
        g.HandleLaserScan(_scan_msg(sc))
    assert e.value.code == -4                                   # more beams than the handle holds
    sc = plate_scan(128, [])
    sc.range_max = sc.range_min
    with pytest.raises(RdetError) as e:
        g.HandleLaserScan(_scan_msg(sc))
    assert e.value.code == -3                                   # malformed (reference: exit(-1))


def test_detector_feeds_the_filter_end_to_end(oracle_lib):
    """LaserScan -> HIP detector -> HIP EKF vs LaserScan -> oracle detector -> oracle EKF."""
    from oracle.binding import OracleEKF
    from reflector_ekf_slam_amd import EKFOptions, ReflectorEKFSLAM, synth
    cfg = synth.SessionConfig("e2e", 40, 12, synth.DIFF, seed=31, speed=1.0, row_spacing=6.0)
    sess = synth.make_session(cfg, max_scans=60)
    rng = np.random.Generator(np.random.PCG64(77))
    g, o = _pair(S2B)
    opt = EKFOptions(init_time=0.0, init_pose=tuple(sess.init_pose), odom_model=cfg.odom_model,
                     linear_velocity_cov=cfg.sigma_v ** 2, angular_velocity_cov=cfg.sigma_w ** 2,
                     observation_cov=cfg.sigma_obs ** 2)
    fg = ReflectorEKFSLAM(opt, max_landmarks=64, auto_grow=False)
    fo = OracleEKF(cfg.odom_model, 0.0, sess.init_pose, opt.linear_velocity_cov, opt.angular_velocity_cov,
                   opt.observation_cov)
    first = True
    for e in range(sess.n_events):
        t = sess.ev_time[e]
        if sess.ev_type[e] == synth.EV_ODOM:
            fg.handle_odometry(t, *sess.odom[e]); fo.handle_odometry(t, *sess.odom[e])
            x, y, th = sess.true_pose[e]
            _feed_odom(g, o, [(t, x, y, math.sin(th / 2), math.cos(th / 2), sess.odom[e][0], 0.0, sess.odom[e][2])])
            continue
        sc = __import__("types").SimpleNamespace(**synth.make_laser_scan(sess.landmarks, sess.true_pose[e], t, rng,
                                                                         n_beams=2880))
        og = g.HandleLaserScan(_scan_msg(sc))
        to, co = o.handle_scan(sc)
        assert og.cloud_.shape == co.shape and np.array_equal(og.cloud_, co)
        if first:
            first = False
            continue
        fg.handle_observation(og.time_, og.cloud_[:64]); fo.handle_observation(to, co[:64])
    assert fg.n == fo.n and fg.n > 3 + 2 * 10
    assert np.abs(fg.mu() - fo.mu()).max() < 1e-9          # identical observations in: the filters agree to FP64 round-off


# ---------------------------------------------------------------------------- 3D detector
def _blob(center, n, spread, rng, intensity=200.0):
    p = rng.normal(0, spread, size=(n, 3)) + np.asarray(center)
    return np.concatenate([p, np.full((n, 1), intensity)], -1)


def _compare3d(cloud, s2b=(0.0, 0.0, 0.0)):
    from oracle.binding import oracle_detect3d
    from reflector_ekf_slam_amd.detect import PointCloudOptions, PointCloudReflectorDetect
    g = PointCloudReflectorDetect(PointCloudOptions(), max_points=65536, sensor_to_base_link=s2b)
    obs = g.HandlePointCloud(3.25, cloud)
    c, m1, m2 = oracle_detect3d(cloud, sensor_to_base_link=s2b)
    assert obs.time_ == 3.25
    assert obs.cloud_.shape == c.shape, (obs.cloud_.shape, c.shape)
    if c.size:
        assert np.abs(obs.cloud_ - c).max() < TOL          # identical clusters, identical order
    return obs


def test_detect3d_micro_cases_match(oracle_lib):
    rng = np.random.default_rng(0)
    dim = _blob((0, 0, 0), 500, 5.0, rng, intensity=20.0)
    a = _blob((3.0, 1.0, 0.5), 40, 0.03, rng)
    b = _blob((-2.0, 4.0, 0.7), 25, 0.03, rng)
    far = _blob((8.0, -6.0, 0.3), 3, 0.01, rng)
    big = _blob((1.0, 1.0, 0.5), 300, 0.055, rng)       # ~220 points survive SOR: above MaxClusterSize
    assert _compare3d(np.concatenate([dim, a, b, far]).astype(np.float32)).cloud_.shape == (2, 2)
    _compare3d(np.concatenate([dim, b]).astype(np.float32))                      # < 31 bright points
    _compare3d(np.concatenate([big, a]).astype(np.float32))                      # oversize cluster dropped
    _compare3d(np.concatenate([a, b]).astype(np.float32), s2b=(1.0, -2.0, math.pi / 2))
    _compare3d(np.zeros((0, 4), np.float32))
    _compare3d(dim.astype(np.float32))                                           # nothing bright at all
    # a long chain (diameter ~ 100 hops of 0.15 m): label propagation must still converge
    chain = np.stack([0.15 * np.arange(120), np.zeros(120), np.zeros(120), np.full(120, 200.0)], -1)
    _compare3d(chain.astype(np.float32))


def test_detect3d_results_do_not_depend_on_the_sorting_grid(oracle_lib):
    """Round 4 sorts the bright points on a 128 x 128 grid whose extent is the PREVIOUS cloud's bounding box (the first cloud's: 64 m
    around the sensor) and prunes the neighbour searches with per-tile boxes.  Nothing of the result may depend on that grid: one
    detector handle is fed clouds whose extents have nothing to do with each other -- 400 m from the sensor (every point clamped into
    border cells), millimetres across, all in one cell, a lone point far outside -- and every answer must be the oracle's, bit for bit."""
    from oracle.binding import oracle_detect3d
    from reflector_ekf_slam_amd.detect import PointCloudOptions, PointCloudReflectorDetect
    rng = np.random.default_rng(11)
    g = PointCloudReflectorDetect(PointCloudOptions(), max_points=65536)

    def scene(centre, scale, n_clusters, rng, per=(8, 60), spread=0.03, outliers=20):
        parts = [_blob(centre, 400, 4.0 * scale, rng, intensity=20.0)]
        for _ in range(n_clusters):
            c = np.asarray(centre) + rng.uniform(-10 * scale, 10 * scale, 3) * np.array([1, 1, 0.05])
            parts.append(_blob(c, int(rng.integers(*per)), spread, rng))
        if outliers:
            o = np.asarray(centre) + rng.uniform(-12 * scale, 12 * scale, (outliers, 3)) * np.array([1, 1, 0.05])
            parts.append(np.concatenate([o, np.full((outliers, 1), 230.0)], -1))
        c = np.concatenate(parts).astype(np.float32)
        return c[rng.permutation(c.shape[0])]                                     # arrival order has nothing to do with space either

    clouds = [scene((0, 0, 0.5), 1.0, 30, rng),
              scene((400.0, -250.0, 0.5), 1.0, 30, rng),                           # far outside the first grid: all clamped
              scene((0.3, 0.2, 0.5), 0.01, 6, rng, spread=0.002, outliers=5),      # centimetres across: one or two cells
              scene((0, 0, 0.5), 3.0, 60, rng),                                    # the grid left by the tiny cloud is useless here
              np.concatenate([scene((5, 5, 0.5), 0.5, 10, rng), np.array([[9000.0, -9000.0, 0.0, 250.0]], np.float32)]),   # a lone far point
              scene((0, 0, 0.5), 1.0, 30, rng)]
    for k, cloud in enumerate(clouds):
        obs = g.HandlePointCloud(1.0 + k, cloud)
        c, m1, m2 = oracle_detect3d(cloud)
        assert obs.cloud_.shape == c.shape, (k, obs.cloud_.shape, c.shape)
        assert np.array_equal(obs.cloud_, c), f"cloud {k}: centres differ from the oracle's"
        assert c.shape[0] >= 3 or k == 2, k                                      # (the centimetre-sized scene is ONE oversize cluster: dropped)
    g.close()


def test_detect3d_non_finite_and_coincident_points(oracle_lib):
    """What a LiDAR driver really sends: no-return points as NaN / inf (bright ones included), NaN intensities, and returns that coincide
    exactly (distance 0 between many neighbours: the 31-NN multiset, the SOR statistics and the 0.2 m clustering all see ties).  The answer
    must be the oracle's -- and the call must come back (no lane may spin on a NaN distance)."""
    from oracle.binding import oracle_detect3d
    from reflector_ekf_slam_amd.detect import PointCloudOptions, PointCloudReflectorDetect
    rng = np.random.default_rng(5)
    parts = [_blob((0, 0, 0), 2000, 5.0, rng, intensity=20.0)]
    parts += [_blob(rng.uniform(-10, 10, 3) * np.array([1, 1, 0.05]), 40, 0.03, rng) for _ in range(20)]
    base = np.concatenate(parts).astype(np.float32)
    g = PointCloudReflectorDetect(PointCloudOptions(), max_points=65536)
    cases = {"nan_xyz": lambda c: c.__setitem__((slice(2100, 2110), slice(0, 3)), np.nan),
             "inf_x": lambda c: c.__setitem__((slice(2100, 2105), 0), np.inf),
             "nan_intensity": lambda c: c.__setitem__((slice(2100, 2110), 3), np.nan),
             "forty_coincident": lambda c: c.__setitem__((slice(2100, 2140), slice(0, 3)), c[2100, :3].copy()),
             "all_coincident": lambda c: c.__setitem__((slice(2000, 2800), slice(0, 3)), c[2100, :3].copy()),
             "clean": lambda c: None}
    for name, mod in cases.items():
        c = base.copy()
        mod(c)
        obs = g.HandlePointCloud(1.0, c)
        oc, _, _ = oracle_detect3d(c)
        assert obs.cloud_.shape == oc.shape and np.array_equal(obs.cloud_, oc), name
        assert np.isfinite(obs.cloud_).all(), name
    g.close()


def test_detect3d_world_clouds_match(oracle_lib):
    from reflector_ekf_slam_amd import synth
    for seed, pose in ((3, (34.4, 34.0, 1.15)), (4, (10.0, 50.0, -0.4))):
        rng = np.random.Generator(np.random.PCG64(seed))
        lms = synth.make_world(synth.C4, rng)
        cloud = synth.make_point_cloud(lms, pose, rng)
        obs = _compare3d(cloud)
        assert obs.cloud_.shape[0] >= 40


def test_detect3d_two_clouds_on_their_way_give_the_synchronous_calls_results(oracle_lib):
    """rdet3d_submit / rdet3d_collect: cloud k + 1 is copied and enqueued while the device is still on cloud k (two input buffers, two sets
    of result slots, of every array in between and two streams: the two chains share the chip).  Ten different clouds -- an empty one among them --
    through the two halves, always two on their way, give bit for bit what HandlePointCloud gives for the same sequence on a twin
    (the sorting grid each cloud inherits from its predecessor included); the misuse cases return errors and leave the pipeline intact."""
    from reflector_ekf_slam_amd import synth
    from reflector_ekf_slam_amd.detect import PointCloudOptions, PointCloudReflectorDetect, RdetError
    rng = np.random.Generator(np.random.PCG64(11))
    lms = synth.make_world(synth.C4, rng)
    clouds = []
    for k in range(10):
        pose = (float(rng.uniform(5, 60)), float(rng.uniform(5, 60)), float(rng.uniform(-3, 3)))
        clouds.append(synth.make_point_cloud(lms, pose, rng, rings=int(rng.choice([8, 16])), n_az=int(rng.choice([900, 1800]))))
    clouds[4] = np.zeros((0, 4), np.float32)
    a = PointCloudReflectorDetect(PointCloudOptions(), max_points=65536)
    b = PointCloudReflectorDetect(PointCloudOptions(), max_points=65536)
    ref = [a.HandlePointCloud(1.0 + k, c) for k, c in enumerate(clouds)]
    with pytest.raises(RdetError):
        b.CollectObservation()                               # nothing submitted
    got = []
    b.SubmitPointCloud(1.0, clouds[0])
    for k in range(1, len(clouds)):
        b.SubmitPointCloud(1.0 + k, clouds[k])               # two on their way
        if k == 3:
            with pytest.raises(RdetError):
                b.SubmitPointCloud(99.0, clouds[0])          # a third one: refused, nothing changes
            with pytest.raises(RdetError):
                b.HandlePointCloud(99.0, clouds[0])          # ... and so is the synchronous call in between
        got.append(b.CollectObservation())
    got.append(b.CollectObservation())
    assert len(got) == len(ref)
    for k, (x, y) in enumerate(zip(got, ref)):
        assert x.time_ == y.time_ and x.cloud_.shape == y.cloud_.shape and np.array_equal(x.cloud_, y.cloud_), k
    assert sum(r.cloud_.shape[0] for r in ref) > 100
    assert np.array_equal(b.HandlePointCloud(50.0, clouds[1]).cloud_, a.HandlePointCloud(50.0, clouds[1]).cloud_)   # (the pipeline is empty again)
    a.close(); b.close()


def test_c4_short_cloud_to_filter_omni(oracle_lib):
    """BASELINE.json configs[3] (shortened): synthetic 3D clouds -> 3D detector -> EKF with the OMNI
    odometry model, HIP path vs oracle path, lock-step."""
    from oracle.binding import OracleEKF, oracle_detect3d
    from reflector_ekf_slam_amd import EKFOptions, ReflectorEKFSLAM, synth
    from reflector_ekf_slam_amd.detect import PointCloudOptions, PointCloudReflectorDetect
    import dataclasses
    cfg = dataclasses.replace(synth.C4, name="C4_short")
    sess = synth.make_session(cfg, max_scans=70)
    rng = np.random.Generator(np.random.PCG64(99))
    det = PointCloudReflectorDetect(PointCloudOptions(), max_points=32768)
    opt = EKFOptions(init_time=0.0, init_pose=tuple(sess.init_pose), odom_model=cfg.odom_model,
                     linear_velocity_cov=cfg.sigma_v ** 2, angular_velocity_cov=cfg.sigma_w ** 2,
                     observation_cov=cfg.sigma_obs ** 2)
    fg = ReflectorEKFSLAM(opt, max_landmarks=cfg.n_landmarks, auto_grow=False)
    fo = OracleEKF(cfg.odom_model, 0.0, sess.init_pose, opt.linear_velocity_cov, opt.angular_velocity_cov,
                   opt.observation_cov)
    first = True
    n_obs = 0
    for e in range(sess.n_events):
        t = sess.ev_time[e]
        if sess.ev_type[e] == synth.EV_ODOM:
            fg.handle_odometry(t, *sess.odom[e]); fo.handle_odometry(t, *sess.odom[e])
            continue
        cloud = synth.make_point_cloud(sess.landmarks, sess.true_pose[e], rng, n_outliers=20)
        og = det.HandlePointCloud(t, cloud)
        co, _, _ = oracle_detect3d(cloud)
        assert og.cloud_.shape == co.shape and np.abs(og.cloud_ - co).max() < TOL
        if first:
            first = False
            continue
        k = min(co.shape[0], 64)
        fg.handle_observation(t, og.cloud_[:k]); fo.handle_observation(t, co[:k])
        n_obs += 1
        sg, so = fg.last_match(), fo.last_match()
        assert np.array_equal(sg.state_obs_match_ids, so[0]) and np.array_equal(sg.new_ids, so[2])
    assert n_obs >= 60 and fg.n == fo.n and fg.n > 3 + 2 * 40
    assert np.abs(fg.mu() - fo.mu()).max() < 1e-4


# ---------------------------------------------------------------- BASELINE.json configs[3] at FULL size
@pytest.fixture(scope="module")
def c4_built():
    """C4 as the reference would run it (src/ros_node.cc:563-625): every scan of the whole session is a synthetic
    XYZI sweep that goes through HandlePointCloud (3D detector, GPU) and then HandleObservationMessage with the OMNI
    odometry model, until the map holds all N = 512 posts (n = 1027).  GPU only: the oracle joins afterwards."""
    from reflector_ekf_slam_amd import EKFOptions, ReflectorEKFSLAM, synth
    from reflector_ekf_slam_amd.detect import PointCloudOptions, PointCloudReflectorDetect
    cfg = synth.C4
    sess = synth.make_session(cfg)
    rng = np.random.Generator(np.random.PCG64(cfg.seed + 7))
    det = PointCloudReflectorDetect(PointCloudOptions(), max_points=32768)
    opt = EKFOptions(init_time=0.0, init_pose=tuple(sess.init_pose), odom_model=cfg.odom_model,
                     linear_velocity_cov=cfg.sigma_v ** 2, angular_velocity_cov=cfg.sigma_w ** 2,
                     observation_cov=cfg.sigma_obs ** 2)
    g = ReflectorEKFSLAM(opt, max_landmarks=cfg.n_landmarks, auto_grow=False)
    first, kmax, scans = True, 0, 0
    for e in range(sess.n_events):
        t = sess.ev_time[e]
        if sess.ev_type[e] == synth.EV_ODOM:
            g.handle_odometry(t, *sess.odom[e])
            continue
        cloud = synth.make_point_cloud(sess.landmarks, sess.true_pose[e], rng, n_outliers=20, **synth.C4_LIDAR)
        ob = det.HandlePointCloud(t, cloud)
        if first:                                   # src/ros_node.cc:566-579 (Q11)
            first = False
            continue
        kmax = max(kmax, ob.cloud_.shape[0])
        g.HandleObservationMessage(ob)
        scans += 1
    assert g.sync_code() == 0                       # no capacity overflow, no singular S along the way
    return cfg, sess, g, det, rng, dict(kmax=kmax, scans=scans)


def test_c4_full_size_properties(c4_built):
    cfg, sess, g, det, rng, info = c4_built
    n = 3 + 2 * cfg.n_landmarks
    # the statistical outlier removal (MeanK = 30) only lets posts with >= ~31 hits through: about 20 per sweep
    assert info["scans"] > 2500 and 16 < info["kmax"] <= 64
    st = g.GetState()
    assert st.mu.shape[0] == n == 1027 and st.sigma.shape == (n, n)
    assert np.isfinite(st.mu).all() and np.isfinite(st.sigma).all()
    scale = np.abs(st.sigma).max()
    assert np.abs(st.sigma - st.sigma.T).max() < 1e-12 * max(scale, 1.0)
    assert np.diag(st.sigma).min() > 0 and -math.pi < st.mu[2] <= math.pi
    # the map is right: every estimated post next to exactly one true post (bijection), pose near the truth
    lm = st.mu[3:].reshape(-1, 2)
    d = np.linalg.norm(lm[:, None] - sess.landmarks[None], axis=-1)
    assert len(set(d.argmin(1).tolist())) == cfg.n_landmarks
    assert d.min(1).max() < 1.0 and np.linalg.norm(st.mu[:2] - sess.true_pose[-1][:2]) < 1.0
    w = np.linalg.eigvalsh(0.5 * (st.sigma[:300, :300] + st.sigma[:300, :300].T))
    assert w.min() > -1e-12 * w.max()


def test_c4_full_size_steps_match_oracle(c4_built, oracle_lib):
    """From the GPU's own n = 1027 state: 16 more sweeps at the parked pose, detector + filter on both paths in
    lock-step (oracle started with oekf set_state): identical centres, identical association lists, same mean."""
    from oracle.binding import OracleEKF, oracle_detect3d
    from reflector_ekf_slam_amd import synth
    cfg, sess, g, det, rng, info = c4_built
    st = g.GetState()
    o = OracleEKF(cfg.odom_model, 0.0, sess.init_pose, cfg.sigma_v ** 2, cfg.sigma_w ** 2, cfg.sigma_obs ** 2)
    vt = sess.odom[np.nonzero(sess.ev_type == synth.EV_ODOM)[0][-1]]
    o.set_state(st.time, st.mu, st.sigma, vt)
    t = st.time
    for k in range(16):
        t += 0.1
        cloud = synth.make_point_cloud(sess.landmarks, sess.true_pose[-1], rng, n_outliers=20, **synth.C4_LIDAR)
        og = det.HandlePointCloud(t, cloud)
        co, _, _ = oracle_detect3d(cloud)
        assert og.cloud_.shape == co.shape and np.abs(og.cloud_ - co).max() < TOL
        g.HandleObservationMessage(og)
        o.handle_observation(t, co)
        sg, so = g.last_match(), o.last_match()
        assert np.array_equal(sg.state_obs_match_ids, so[0]) and np.array_equal(sg.new_ids, so[2])
        assert sg.state_obs_match_ids.shape[0] >= 8 and sg.new_ids.size == 0
        assert np.abs(g.mu() - o.mu()).max() < 1e-9
    mo, Po = o.state()
    assert np.abs(g.GetState().sigma - Po).max() < 1e-12
    assert g.sync_code() == 0
