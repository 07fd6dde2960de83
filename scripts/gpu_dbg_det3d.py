"""In-kernel timeline of k3_knn (per workgroup: start, own tiles swept, tiles selected, sweep done, merged) from a
-DRDET_DEBUG_MARKS build of librdet.so:
  make -C reflector_ekf_slam_amd/csrc -B ../librdet.so HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -DRDET_DEBUG_MARKS"
GPU box: python scripts/gpu_dbg_det3d.py [rings]   (marks: wall_clock64, 10 ns ticks)"""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import synth
from reflector_ekf_slam_amd.detect import PointCloudReflectorDetect, PointCloudOptions

rings = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.Generator(np.random.PCG64(7))
lms = synth.make_world(synth.C4, rng)
pose = (float(lms[:, 0].mean()), float(lms[:, 1].mean()), 0.3)
cloud = synth.make_point_cloud(lms, pose, rng, rings=rings, n_az=1800)
g = PointCloudReflectorDetect(PointCloudOptions(), max_points=65536)
for _ in range(5):
    g.HandlePointCloud(1.0, cloud)
m = np.zeros((2048, 8), np.uint64)
g._L.rdet3d_debug_marks.argtypes = [C.c_void_p, C.c_void_p]
g._L.rdet3d_debug_marks(g._h, m.ctypes.data)
m = m.astype(np.int64)
live = m[:, 0] > 0
t0 = m[live, 0].min()
print("workgroups", int(live.sum()))
print("  wg  start  own   select  sweep  merge  end(us)   listed")
rows = []
for b in np.nonzero(live)[0]:
    r = m[b]
    rows.append((b, (r[0] - t0) / 100, (r[1] - r[0]) / 100, (r[2] - r[1]) / 100, (r[3] - r[2]) / 100, (r[4] - r[3]) / 100, (r[4] - t0) / 100, int(r[6])))
rows.sort(key=lambda x: -x[6])
for r in rows[:12] + rows[-4:]:
    print("%4d %6.2f %6.2f %6.2f %6.2f %6.2f %7.2f %6d" % r)
a = np.array([r[1:] for r in rows])
print("mean", np.round(a.mean(0), 2))
