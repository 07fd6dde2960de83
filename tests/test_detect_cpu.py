"""CPU suite: pins the 2D detector oracle (oracle/detect2d_oracle.c) with hand-checkable scans.
The reference (laser_reflector_detect.cc) has no tests of its own -- parity unpinned -- so each
case below encodes one documented behaviour of its state machine."""
import math

import numpy as np
import pytest

from tests.detect_cases import S2B, beams_for_width, odom_stream, plate_scan, world_scan


def _oracle(**kw):
    from oracle.binding import OracleDetect2D
    return OracleDetect2D(sensor_to_base_link=kw.pop("s2b", (0.0, 0.0, 0.0)), **kw)


def test_single_plate_centre_and_width_gate(oracle_lib):
    n = 720
    rng_ = 5.0
    nb = beams_for_width(0.18, rng_, n)                       # ~0.18 m wide: accepted
    sc = plate_scan(n, [(100, nb, rng_, 200.0), (300, 3 * nb, rng_, 200.0)])   # second one ~0.57 m: rejected
    t, c = _oracle().handle_scan(sc)
    assert t == sc.stamp and c.shape == (1, 2)
    inc = 2 * math.pi / n
    mid = -math.pi + inc * (100 + (nb - 1) / 2.0)
    # centroid of points on an arc of radius 5 around the mid bearing
    assert abs(math.atan2(c[0, 1], c[0, 0]) - mid) < 1e-4
    assert abs(np.hypot(*c[0]) - rng_) < 2e-3


def test_gap_is_bridged_only_under_the_three_conditions(oracle_lib):
    n, rng_ = 720, 5.0
    nb = beams_for_width(0.18, rng_, n)                       # 5 beams
    base = plate_scan(n, [(100, nb, rng_, 200.0)])
    one = _oracle().handle_scan(base)[1]
    # a dim beam in the middle (gap of 1): still ONE reflector, the dim beam's point is included (:111-138)
    sc = plate_scan(n, [(100, nb, rng_, 200.0)])
    sc.intensities[102] = 50.0
    c = _oracle().handle_scan(sc)[1]
    assert c.shape == (1, 2) and np.abs(c - one).max() < 1e-6
    # the dim beam is at a very different range: no bridge -> run splits into 2+2 beams, both too short
    sc.ranges[102] = 9.0
    sc.ranges[103] = 5.4                                     # |r_i - r_last| = 0.4 >= 0.3
    assert _oracle().handle_scan(sc)[1].shape[0] == 0
    # an inf gap beam is skipped, not averaged in (:120-121)
    sc2 = plate_scan(n, [(100, nb, rng_, 200.0)])
    sc2.intensities[102] = 50.0
    sc2.ranges[102] = np.inf
    c2 = _oracle().handle_scan(sc2)[1]
    assert c2.shape == (1, 2) and np.isfinite(c2).all()


def test_wraparound_union_of_first_and_last_run(oracle_lib):
    n, rng_ = 720, 5.0
    nb = beams_for_width(0.18, rng_, n)                       # 5 beams
    # 2 beams at the end + 3 at the start of the scan: one plate across the +-pi seam (:188-195)
    sc = plate_scan(n, [(n - 2, 2, rng_, 200.0), (0, 3, rng_, 200.0), (200, nb, rng_, 200.0)])
    c = _oracle().handle_scan(sc)[1]
    assert c.shape == (2, 2)
    # cluster 0 is the seam plate: its centre sits at bearing ~ -pi/+pi
    assert abs(abs(math.atan2(c[0, 1], c[0, 0])) - math.pi) < 0.02
    assert abs(np.hypot(*c[0]) - rng_) < 5e-3


def test_circle_scan_first_run_from_beam0_is_accepted_unconditionally(oracle_lib):
    n, rng_ = 720, 5.0
    sc = plate_scan(n, [(0, 2, rng_, 200.0), (200, beams_for_width(0.18, rng_, n), rng_, 200.0)])
    c = _oracle().handle_scan(sc)[1]
    # the 2-beam run starting at beam 0 (0.04 m wide) passes only through the circle-scan clause (:151)
    assert c.shape == (2, 2)


def test_no_bright_beam_is_an_empty_observation_and_bad_scans_are_errors(oracle_lib):
    sc = plate_scan(360, [])
    t, c = _oracle().handle_scan(sc)
    assert c.shape == (0, 2)                                  # Q13: defined, the reference has UB here
    sc.range_max = sc.range_min
    with pytest.raises(ValueError):
        _oracle().handle_scan(sc)                             # reference: LOG(ERROR) + exit(-1), :27-32


def test_deskew_moves_points_and_keeps_the_last_point_fixed(oracle_lib):
    sc, _ = world_scan()
    still = _oracle(s2b=S2B)
    t0, c0 = still.handle_scan(sc)
    r0 = still.returns()
    mov = _oracle(s2b=S2B)
    for o in odom_stream(sc.stamp - 0.3, sc.stamp + 0.05):
        mov.handle_odometry(*o)
    t1, c1 = mov.handle_scan(sc)
    r1 = mov.returns()
    assert c0.shape == c1.shape and c0.shape[0] >= 16
    assert np.abs(r1[-1] - r0[-1]).max() < 1e-5               # the scan-end frame: last point unchanged
    assert 0.01 < np.abs(r1[0] - r0[0]).max() < 2.0           # first point (a 24 m wall hit) moved by ~v*dt + r*w*dt
    assert np.abs(c1 - c0).max() > 1e-3


def test_world_scan_detects_the_true_reflectors(oracle_lib):
    sc, lms = world_scan()
    c = _oracle(s2b=S2B).handle_scan(sc)[1]
    pose = (16.0, 17.7, 0.6)
    rel = lms - np.array(pose[:2])
    cc, ss = math.cos(pose[2]), math.sin(pose[2])
    gt = np.stack([cc * rel[:, 0] + ss * rel[:, 1], -ss * rel[:, 0] + cc * rel[:, 1]], -1)
    d = np.linalg.norm(gt[None] - c[:, None], axis=-1).min(1)
    assert c.shape[0] >= 20 and d.max() < 0.02


# ---------------------------------------------------------------------------- 3D detector oracle
def _blob(center, n, spread, rng, intensity=200.0):
    p = rng.normal(0, spread, size=(n, 3)) + np.asarray(center)
    return np.concatenate([p, np.full((n, 1), intensity)], -1)


def test_detect3d_micro_cases(oracle_lib):
    """PCL semantics restated in oracle/detect3d_oracle.c (parity unpinned: PCL absent, no reference tests)."""
    from oracle.binding import oracle_detect3d
    rng = np.random.default_rng(0)
    dim = _blob((0, 0, 0), 500, 5.0, rng, intensity=20.0)                 # never passes the intensity gate (:33)
    a = _blob((3.0, 1.0, 0.5), 40, 0.03, rng)
    b = _blob((-2.0, 4.0, 0.7), 25, 0.03, rng)
    far = _blob((8.0, -6.0, 0.3), 3, 0.01, rng)                           # 3 points: below MinClusterSize 4
    cloud = np.concatenate([dim, a, b, far]).astype(np.float32)
    c, m1, m2 = oracle_detect3d(cloud)
    assert m1 == 68 and m2 <= 68
    assert c.shape == (2, 2)                                               # largest first
    assert np.abs(c[0] - [3.0, 1.0]).max() < 0.02 and np.abs(c[1] - [-2.0, 4.0]).max() < 0.02
    # fewer than MeanK+1 = 31 bright points: every k-NN search "fails", nothing is removed
    c2, m1, m2 = oracle_detect3d(np.concatenate([dim, b]).astype(np.float32))
    assert m1 == 25 and m2 == 25 and c2.shape == (1, 2)
    # a cluster above MaxClusterSize 160 is dropped as a whole (:71)
    big = _blob((1.0, 1.0, 0.5), 300, 0.055, rng)       # ~220 points survive SOR: above MaxClusterSize
    c3, _, _ = oracle_detect3d(np.concatenate([big, a]).astype(np.float32))
    assert c3.shape == (1, 2) and np.abs(c3[0] - [3.0, 1.0]).max() < 0.02
    # sensor_to_base_link is applied as a Rigid2f (:96); z never matters (Q16)
    c4, _, _ = oracle_detect3d(np.concatenate([a, b]).astype(np.float32), sensor_to_base_link=(1.0, -2.0, math.pi / 2))
    assert np.abs(c4[0] - [1.0 - 1.0, -2.0 + 3.0]).max() < 0.02
    assert oracle_detect3d(np.zeros((0, 4), np.float32))[0].shape == (0, 2)


def test_detect3d_world_cloud(oracle_lib):
    from oracle.binding import oracle_detect3d
    from reflector_ekf_slam_amd import synth
    rng = np.random.Generator(np.random.PCG64(3))
    lms = synth.make_world(synth.C4, rng)
    pose = (34.4, 34.0, 1.15)
    cloud = synth.make_point_cloud(lms, pose, rng)
    c, m1, m2 = oracle_detect3d(cloud)
    rel = lms - np.array(pose[:2])
    cc, ss = math.cos(pose[2]), math.sin(pose[2])
    gt = np.stack([cc * rel[:, 0] + ss * rel[:, 1], -ss * rel[:, 0] + cc * rel[:, 1]], -1)
    d = np.linalg.norm(gt[None] - c[:, None], axis=-1).min(1)
    assert c.shape[0] >= 40 and d.max() < 0.05 and m2 < m1        # the planted outliers are gone


def test_device_sincosf_restatement_equals_the_host_libm_bit_for_bit(tmp_path):
    """csrc/glibc_sincosf.h is what the 2D detector's kernels evaluate float32 sin / cos with: glibc's own algorithm (Arm Optimized
    Routines), FMA build.  Compiled for the HOST here and compared with the host's sinf / cosf on 6e6 arguments up to |x| = 110:
    zero mismatches, which is what makes the detector's centres bit-identical to the CPU oracle's (tests/test_detect_gpu.py).
    Needs a CPU with FMA (glibc's ifunc then runs the same fused sequence); skipped otherwise."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    try:
        if " fma " not in open("/proc/cpuinfo").read().replace("\n", " "):
            pytest.skip("host CPU without FMA: glibc takes its non-fused path")
    except OSError:
        pass
    exe = str(tmp_path / "sincosf_check")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(root, "reflector_ekf_slam_amd", "csrc"),
                    os.path.join(root, "tests", "cpp", "sincosf_check.cpp"), "-o", exe, "-lm"], check=True)
    bad_s, bad_c, n = (int(v) for v in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split())
    assert n == 6000000 and bad_s == 0 and bad_c == 0
