"""Call latency of the 3D detector alone (one cloud shape; for rocprofv3 runs that should hold nothing but its kernels).
Usage (GPU box): python scripts/gpu_det3d_only.py [reps] [rings]"""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import synth
from reflector_ekf_slam_amd.detect import PointCloudReflectorDetect, PointCloudOptions
from oracle.binding import oracle_detect3d

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rings = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.Generator(np.random.PCG64(7))
for _ in range(3):
    synth.make_world(synth.C2 if _ < 2 else synth.C3, rng)       # (the generator state gpu_bench_detectors.py has when it reaches the clouds)
lms = synth.make_world(synth.C4, rng)
pose = (float(lms[:, 0].mean()), float(lms[:, 1].mean()), 0.3)
cloud = synth.make_point_cloud(lms, pose, rng, rings=rings, n_az=1800)
g3 = PointCloudReflectorDetect(PointCloudOptions(), max_points=65536)
obs = g3.HandlePointCloud(1.0, cloud)
co, m1, m2 = oracle_detect3d(cloud)
same = co.shape == obs.cloud_.shape and bool((obs.cloud_ == co).all())
for _ in range(20):
    g3.HandlePointCloud(1.0, cloud)
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    g3.HandlePointCloud(1.0, cloud)
    ts.append((time.perf_counter() - t0) * 1e6)
ts = np.array(ts)
print(json.dumps({"points": int(cloud.shape[0]), "after_gate": int(m1), "after_sor": int(m2), "centres": int(obs.cloud_.shape[0]),
                  "identical_to_oracle": same, "call_us": {"median": round(float(np.median(ts)), 1), "p10": round(float(np.percentile(ts, 10)), 1),
                                                            "p90": round(float(np.percentile(ts, 90)), 1), "mean": round(float(ts.mean()), 1)}}))
