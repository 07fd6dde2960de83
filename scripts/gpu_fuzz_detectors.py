"""Randomised soak of the 2D / 3D reflector detectors against the CPU oracle: world scans from random poses with random
dropouts (dim beams inside plates, inf / out-of-range returns, range steps), with and without odometry (de-skew), random
beam counts; 3D clouds of random blobs (sizes around the cluster-size gates, touching pairs, outliers).
GPU box: python scripts/gpu_fuzz_detectors.py [n2d] [n3d] -> one summary JSON line (+ one line per failing seed)."""
import json, math, sys
sys.path.insert(0, ".")
import numpy as np
from types import SimpleNamespace as NS
from reflector_ekf_slam_amd import OdometryData, synth
from reflector_ekf_slam_amd.detect import (LaserReflectorDetect, LaserScan, PointCloudOptions, PointCloudReflectorDetect,
                                           ReflectorDetectOptions)
from oracle.binding import OracleDetect2D, oracle_detect3d
from tests.detect_cases import S2B, odom_stream

n2d = int(sys.argv[1]) if len(sys.argv) > 1 else 120
n3d = int(sys.argv[2]) if len(sys.argv) > 2 else 60
TOL = 1e-5
fails = []
worst2 = worst3 = worst_ret2 = 0.0
refl2 = refl3 = exact2 = far2 = 0
for seed in range(n2d):
    rng = np.random.Generator(np.random.PCG64(5000 + seed))
    lms = synth.make_world(synth.C2, rng)
    pose = (float(rng.uniform(2, 32)), float(rng.uniform(2, 32)), float(rng.uniform(-math.pi, math.pi)))
    nb = int(rng.choice([360, 720, 1440, 2880, 3600, 5000]))
    sc = NS(**synth.make_laser_scan(lms, pose, 10.0 + seed, rng, n_beams=nb))
    bright = np.flatnonzero(sc.intensities > 100.0)
    if bright.size:
        for b in rng.choice(bright, size=min(bright.size, int(rng.integers(0, 12))), replace=False):
            kind = int(rng.integers(0, 4))
            if kind == 0: sc.intensities[b] = 50.0                        # dim beam inside a plate (gap bridging)
            elif kind == 1: sc.ranges[b] = np.inf
            elif kind == 2: sc.ranges[b] += np.float32(rng.choice([0.25, 0.35]))   # range step around the 0.3 m gate
            else: sc.ranges[b] = np.float32(100.0)                        # beyond the message's range_max
    if rng.random() < 0.3:                                                # a plate across the seam
        k = int(rng.integers(2, 6))
        sc.ranges[:k] = 3.0; sc.ranges[-k:] = 3.0; sc.intensities[:k] = 200.0; sc.intensities[-k:] = 200.0
    o2 = OracleDetect2D(sensor_to_base_link=S2B)
    g = LaserReflectorDetect(ReflectorDetectOptions(), max_beams=8192, sensor_to_base_link=S2B)   # fresh odometry deque per scan
    if rng.random() < 0.6:
        # every third odometer runs far from its origin (|coordinates| > 128 m: the reference's float32 Rigid2f then carries
        # 1.5e-5 m per ulp through the de-skew -- both sides must still agree bit for bit)
        ox, oy = (float(rng.uniform(150, 400)), float(rng.uniform(-400, -150))) if seed % 3 == 0 else (0.0, 0.0)
        far2 += seed % 3 == 0
        for (t, px, py, qz, qw, vx, vy, wz) in odom_stream(sc.stamp - 0.3, sc.stamp + 0.05, v=float(rng.uniform(0, 2)), w=float(rng.uniform(-1, 1))):
            g.HandleOdometryData(OdometryData(t, (vx, vy, 0.0), (0.0, 0.0, wz), (px + ox, py + oy, 0.0), (qw, 0.0, 0.0, qz)))
            o2.handle_odometry(t, px + ox, py + oy, qz, qw, vx, vy, wz)
    obs = g.HandleLaserScan(LaserScan(sc.stamp, sc.angle_min, sc.angle_max, sc.angle_increment, sc.scan_time, sc.range_min, sc.range_max,
                                      sc.ranges, sc.intensities))
    t, c = o2.handle_scan(sc)
    rg, ro = g.GetRangeData().returns, o2.returns()
    ok = obs.cloud_.shape == c.shape and rg.shape == ro.shape
    if ok and c.size:
        d = float(np.abs(obs.cloud_ - c).max()); worst2 = max(worst2, d); ok = d < TOL
    if ok and ro.size:
        dr = float(np.abs(rg - ro).max()); worst_ret2 = max(worst_ret2, dr)
        ok = dr < 2e-5 * max(1.0, float(np.abs(ro).max()))
    exact2 += bool(ok and np.array_equal(obs.cloud_, c) and np.array_equal(rg, ro))
    refl2 += c.shape[0]
    if not ok:
        fails.append({"kind": "2d", "seed": seed, "beams": nb, "gpu": list(obs.cloud_.shape), "oracle": list(c.shape)})

g3 = PointCloudReflectorDetect(PointCloudOptions(), max_points=65536, sensor_to_base_link=(0.2, -0.1, 0.3))
g3l = PointCloudReflectorDetect(PointCloudOptions(), max_points=65536, sensor_to_base_link=(0.2, -0.1, 0.3))
g3l.debug_set_path(1)                                   # the long chain whatever the size (g3: the short one up to 5120 survivors, by the previous cloud's count)
exact3 = 0
for seed in range(n3d):
    rng = np.random.default_rng(7000 + seed)
    parts = []
    ndim = int(rng.integers(0, 4000))
    parts.append(np.concatenate([rng.normal(0, 6.0, (ndim, 3)), rng.uniform(0, 90, (ndim, 1))], -1))          # dim background
    for _ in range(int(rng.integers(0, 40)) if seed % 9 != 8 else int(rng.integers(60, 160))):      # (every ninth cloud: 4 - 12 k bright points)
        centre = rng.uniform(-15, 15, 3) * [1, 1, 0.05]
        n = int(rng.choice([2, 3, 4, 5, 30, 31, 32, 60, 159, 160, 161, 200]))
        spread = float(rng.choice([0.01, 0.03, 0.06, 0.12]))
        parts.append(np.concatenate([rng.normal(0, spread, (n, 3)) + centre, np.full((n, 1), 200.0)], -1))
        if rng.random() < 0.3:                                                                                # a neighbour near the 0.2 m tolerance
            parts.append(np.concatenate([rng.normal(0, 0.02, (12, 3)) + centre + [float(rng.choice([0.15, 0.2, 0.25, 0.4])), 0, 0],
                                         np.full((12, 1), 200.0)], -1))
    cloud = np.concatenate(parts).astype(np.float32)
    cloud = cloud[rng.permutation(cloud.shape[0])]
    obs = g3.HandlePointCloud(1.0 + seed, cloud)
    obl = g3l.HandlePointCloud(1.0 + seed, cloud)
    c, m1, m2 = oracle_detect3d(cloud, sensor_to_base_link=(0.2, -0.1, 0.3))
    ok = obs.cloud_.shape == c.shape and obl.cloud_.shape == c.shape
    if ok and c.size:
        d = max(float(np.abs(obs.cloud_ - c).max()), float(np.abs(obl.cloud_ - c).max())); worst3 = max(worst3, d); ok = d < TOL
    exact3 += bool(ok and np.array_equal(obs.cloud_, c) and np.array_equal(obl.cloud_, c))
    refl3 += c.shape[0]
    if not ok:
        fails.append({"kind": "3d", "seed": seed, "points": int(cloud.shape[0]), "gpu": list(obs.cloud_.shape), "oracle": list(c.shape)})
for f in fails:
    print(json.dumps(f))
print(json.dumps({"summary": True, "scans_2d": n2d, "clouds_3d": n3d, "failed": len(fails), "reflectors_2d": refl2, "clusters_3d": refl3,
                  "worst_centre_diff_2d_m": worst2, "worst_return_diff_2d_m": worst_ret2, "scans_2d_bit_identical": exact2,
                  "scans_2d_with_far_odometry": int(far2), "worst_centre_diff_3d_m": worst3,
                  "clouds_3d_bit_identical_both_chains": exact3, "clouds_3d_short_front_end_and_sent_again": list(g3.debug_path_counts())}))
