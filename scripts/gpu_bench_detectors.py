"""Latency of the two detectors through the C ABI (host buffers in, centres out: PCIe inclusive, one
synchronising call per scan like the reference's callback) next to the CPU oracle on the same inputs.
Usage (GPU box): python scripts/gpu_bench_detectors.py [reps]   -> one JSON line per case."""
import json, math, sys, time
sys.path.insert(0, ".")
from types import SimpleNamespace as NS
import numpy as np
from reflector_ekf_slam_amd import synth
from reflector_ekf_slam_amd.detect import (LaserReflectorDetect, PointCloudReflectorDetect, ReflectorDetectOptions,
                                           PointCloudOptions)
from reflector_ekf_slam_amd import OdometryData
from oracle.binding import OracleDetect2D, oracle_detect3d

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
S2B = (0.13686, 0.0, 0.0)


def timeit(f, n):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e6


rng = np.random.Generator(np.random.PCG64(7))
for name, cfg, beams in (("2d_1440", synth.C2, 1440), ("2d_3600", synth.C2, 3600), ("2d_3600_C3world", synth.C3, 3600)):
    lms = synth.make_world(cfg, rng)
    pose = (float(lms[:, 0].mean()), float(lms[:, 1].mean()), 0.6)
    scan = NS(**synth.make_laser_scan(lms, pose, 10.0, rng, n_beams=beams))
    g = LaserReflectorDetect(ReflectorDetectOptions(), sensor_to_base_link=S2B)
    o = OracleDetect2D(sensor_to_base_link=S2B)
    for k in range(30):
        t = 9.5 + 0.02 * k
        g.HandleOdometryData(OdometryData(time=t, position=(0.5 * t, 0.0, 0.0), orientation=(1.0, 0.0, 0.0, 0.0),
                                          linear_velocity=(0.5, 0.0, 0.0), angular_velocity=(0.0, 0.0, 0.1)))
        o.handle_odometry(t, 0.5 * t, 0.0, 0.0, 1.0, 0.5, 0.0, 0.1)
    obs = g.HandleLaserScan(scan)
    _, co = o.handle_scan(scan)
    err = float(np.abs(obs.cloud_ - co).max()) if co.shape == obs.cloud_.shape and co.size else float("nan")
    print(json.dumps({"case": name, "beams": beams, "centres": int(obs.cloud_.shape[0]), "max_abs_diff_vs_oracle_m": err,
                      "gpu_call_us": round(timeit(lambda: g.HandleLaserScan(scan), reps), 1),
                      "cpu_oracle_us": round(timeit(lambda: o.handle_scan(scan), max(reps // 4, 10)), 1)}))

for name, cfg, rings, n_az in (("3d_16x1800", synth.C4, 16, 1800), ("3d_32x1800", synth.C4, 32, 1800)):
    lms = synth.make_world(cfg, rng)
    pose = (float(lms[:, 0].mean()), float(lms[:, 1].mean()), 0.3)
    cloud = synth.make_point_cloud(lms, pose, rng, rings=rings, n_az=n_az)
    g3 = PointCloudReflectorDetect(PointCloudOptions(), max_points=max(65536, cloud.shape[0]))
    obs = g3.HandlePointCloud(1.0, cloud)
    co, m1, m2 = oracle_detect3d(cloud)
    err = float(np.abs(obs.cloud_ - co).max()) if co.shape == obs.cloud_.shape and co.size else float("nan")
    print(json.dumps({"case": name, "points": int(cloud.shape[0]), "after_intensity_gate": int(m1), "after_sor": int(m2),
                      "centres": int(obs.cloud_.shape[0]), "max_abs_diff_vs_oracle_m": err,
                      "gpu_call_us": round(timeit(lambda: g3.HandlePointCloud(1.0, cloud), reps), 1),
                      "cpu_oracle_us": round(timeit(lambda: oracle_detect3d(cloud), 5), 1)}))
