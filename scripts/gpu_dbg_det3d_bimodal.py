"""Why does bench.py's detectors leg see a bimodal 3D call (75 / 110 us)?  The same loop, with and without EKF work in the process before it, with the
distribution printed.  GPU box: python scripts/gpu_dbg_det3d_bimodal.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
if "torch" in sys.argv:
    import torch
    torch.zeros(8, device="cuda").sum().item()
    torch.cuda.synchronize()
from reflector_ekf_slam_amd import synth
from reflector_ekf_slam_amd.detect import PointCloudReflectorDetect, PointCloudOptions


def dist(ts):
    return "median %.1f  p10 %.1f  p25 %.1f  p75 %.1f  p90 %.1f  p99 %.1f  mean %.1f" % (np.median(ts), *np.percentile(ts, [10, 25, 75, 90, 99]), ts.mean())


def loop(g, cloud, n):
    g.HandlePointCloud(1.0, cloud); g.HandlePointCloud(1.0, cloud)
    ts = np.zeros(n)
    for k in range(n):
        t0 = time.perf_counter()
        g.HandlePointCloud(1.0, cloud)
        ts[k] = 1e6 * (time.perf_counter() - t0)
    return ts


rng = np.random.Generator(np.random.PCG64(7))
lms = synth.make_world(synth.C2, rng)
lms = synth.make_world(synth.C4, rng)
pose = (float(lms[:, 0].mean()), float(lms[:, 1].mean()), 0.3)
cloud = synth.make_point_cloud(lms, pose, rng, rings=16, n_az=1800)
for mode in (0, 1):
    g = PointCloudReflectorDetect(PointCloudOptions(), max_points=65536)
    g.debug_set_path(mode)
    for rep in range(3):
        ts = loop(g, cloud, 200)
        print("mode %d fresh handle, pass %d: %s" % (mode, rep, dist(ts)))
    g.close()
# a 2D detector handle created and destroyed before (as bench.py does), then the 3D loop
from types import SimpleNamespace as NS
from reflector_ekf_slam_amd.detect import LaserReflectorDetect, ReflectorDetectOptions
scan = NS(**synth.make_laser_scan(lms, pose, 10.0, rng, n_beams=3600))
g2 = LaserReflectorDetect(ReflectorDetectOptions(), sensor_to_base_link=(0.1, 0, 0))
for _ in range(100):
    g2.HandleLaserScan(scan)
g2.close()
for mode in (0, 1):
    g = PointCloudReflectorDetect(PointCloudOptions(), max_points=65536)
    g.debug_set_path(mode)
    ts = loop(g, cloud, 200)
    print("mode %d after a 2D handle: %s" % (mode, dist(ts)))
    ts = loop(g, cloud, 200)
    print("mode %d after a 2D handle, again: %s" % (mode, dist(ts)))
    g.close()
# an EKF session in the process before (bench.py's detectors leg runs behind the filter legs)
if "ekf" in sys.argv:
    from reflector_ekf_slam_amd import ReflectorEKFSLAM, EKFOptions
    import bench
    f = bench.gpu_filter_factory(synth.C3, 0) if hasattr(bench, "gpu_filter_factory") else None
for mode in (0, 1):
    g = PointCloudReflectorDetect(PointCloudOptions(), max_points=65536)
    g.debug_set_path(mode)
    ts = loop(g, cloud, 200)
    print("mode %d at the end: %s" % (mode, dist(ts)))
    ts = loop(g, cloud, 200)
    print("mode %d at the end, again: %s" % (mode, dist(ts)))
    hist, edges = np.histogram(ts, bins=[0, 70, 75, 80, 85, 90, 100, 110, 120, 150, 1000])
    print("   histogram (us):", dict(zip(["<70", "70-75", "75-80", "80-85", "85-90", "90-100", "100-110", "110-120", "120-150", ">150"], hist.tolist())))
    g.close()
