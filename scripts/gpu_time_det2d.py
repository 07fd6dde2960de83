"""Cost of one rdet2d_handle_scan call at the C ABI (arguments prepared once, so the loop is the ctypes call alone):
median / p99 over `reps` calls for a 1440- and a 3600-beam scan.  GPU box: python scripts/gpu_time_det2d.py [reps]"""
import ctypes as C, json, sys, time
sys.path.insert(0, ".")
from types import SimpleNamespace as NS
import numpy as np
from reflector_ekf_slam_amd import OdometryData, synth
from reflector_ekf_slam_amd.detect import LaserReflectorDetect, ReflectorDetectOptions, MAX_CENTERS

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.Generator(np.random.PCG64(7))
for beams in (1440, 3600):
    lms = synth.make_world(synth.C2, rng)
    pose = (float(lms[:, 0].mean()), float(lms[:, 1].mean()), 0.6)
    scan = NS(**synth.make_laser_scan(lms, pose, 10.0, rng, n_beams=beams))
    g = LaserReflectorDetect(ReflectorDetectOptions(), sensor_to_base_link=(0.13686, 0.0, 0.0))
    for k in range(30):
        t = 9.5 + 0.02 * k
        g.HandleOdometryData(OdometryData(time=t, position=(0.5 * t, 0.0, 0.0), orientation=(1.0, 0.0, 0.0, 0.0),
                                          linear_velocity=(0.5, 0.0, 0.0), angular_velocity=(0.0, 0.0, 0.1)))
    ranges = np.ascontiguousarray(scan.ranges, np.float32); inten = np.ascontiguousarray(scan.intensities, np.float32)
    centers = np.zeros((MAX_CENTERS, 2), np.float32)
    K = C.c_int(); t = C.c_double()
    f = g._L.rdet2d_handle_scan
    args = (g._h, float(scan.stamp), scan.angle_min, scan.angle_max, scan.angle_increment, scan.scan_time, scan.range_min,
            scan.range_max, ranges.ctypes.data_as(C.c_void_p), inten.ctypes.data_as(C.c_void_p), beams,
            centers.ctypes.data_as(C.c_void_p), MAX_CENTERS, C.byref(K), C.byref(t))
    for _ in range(50):
        assert f(*args) == 0
    us = np.empty(reps)
    for i in range(reps):
        t0 = time.perf_counter_ns()
        f(*args)
        us[i] = (time.perf_counter_ns() - t0) * 1e-3
    print(json.dumps({"beams": beams, "centres": K.value, "call_us_median": round(float(np.median(us)), 1),
                      "call_us_p99": round(float(np.percentile(us, 99)), 1), "call_us_min": round(float(us.min()), 1), "reps": reps}))
