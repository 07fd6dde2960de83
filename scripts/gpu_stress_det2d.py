"""Race hunt for the one-launch 2D detector (in-launch hand-over between workgroups, results polled from pinned memory while
the kernel is still running): N calls alternating between a few different scans through ONE handle; every call's centres and
point cloud must equal that scan's first result bit for bit.  GPU box: python scripts/gpu_stress_det2d.py [calls]"""
import json, sys, time
sys.path.insert(0, ".")
from types import SimpleNamespace as NS
import numpy as np
from reflector_ekf_slam_amd import OdometryData, synth
from reflector_ekf_slam_amd.detect import LaserReflectorDetect, LaserScan, ReflectorDetectOptions

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.Generator(np.random.PCG64(11))
lms = synth.make_world(synth.C2, rng)
scans = []
for k, nb in enumerate((3600, 1440, 3600, 720, 2880, 5000)):
    pose = (float(rng.uniform(4, 30)), float(rng.uniform(4, 30)), float(rng.uniform(-3, 3)))
    sc = NS(**synth.make_laser_scan(lms, pose, 10.0, rng, n_beams=nb))
    scans.append(LaserScan(sc.stamp, sc.angle_min, sc.angle_max, sc.angle_increment, sc.scan_time, sc.range_min, sc.range_max, sc.ranges, sc.intensities))
g = LaserReflectorDetect(ReflectorDetectOptions(), max_beams=8192, sensor_to_base_link=(0.13686, 0.0, 0.0))
for k in range(30):
    t = 9.5 + 0.02 * k
    g.HandleOdometryData(OdometryData(time=t, position=(0.5 * t, 0.0, 0.0), orientation=(1.0, 0.0, 0.0, 0.0),
                                      linear_velocity=(0.5, 0.0, 0.0), angular_velocity=(0.0, 0.0, 0.1)))
ref = []
for sc in scans:
    o = g.HandleLaserScan(sc)
    ref.append((o.cloud_.copy(), g.GetRangeData().returns.copy()))
bad = 0
t0 = time.time()
order = rng.integers(0, len(scans), size=calls)
for n, k in enumerate(order):
    o = g.HandleLaserScan(scans[k])
    ok = o.cloud_.shape == ref[k][0].shape and np.array_equal(o.cloud_, ref[k][0])
    if ok and n % 7 == 0:
        r = g.GetRangeData().returns
        ok = r.shape == ref[k][1].shape and np.array_equal(r, ref[k][1])
    if not ok:
        bad += 1
        if bad < 5:
            print(json.dumps({"call": int(n), "scan": int(k), "got": list(o.cloud_.shape), "want": list(ref[k][0].shape)}))
print(json.dumps({"calls": calls, "mismatches": bad, "reflectors": [int(r[0].shape[0]) for r in ref], "seconds": round(time.time() - t0, 1)}))
