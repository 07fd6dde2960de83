"""Timelines from a -DRDET_DEBUG_MARKS build of librdet.so:
  make -C reflector_ekf_slam_amd/csrc -B ../librdet.so HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -DRDET_DEBUG_MARKS"
GPU box: RDET3_HOST_MARKS=1 python scripts/gpu_dbg_det3d.py [rings]
 * host side of rdet3d_handle_cloud on stderr (stream idle, cloud written, kernels enqueued, head seen, centres out);
 * k3_clusters per workgroup (wave 0): start, roots gated, ranked, members gathered, end (wall_clock64, 10 ns ticks)."""
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
for _ in range(8):
    g.HandlePointCloud(1.0, cloud)
m = np.zeros((2048, 8), np.uint64)
g._L.rdet3d_debug_marks.argtypes = [C.c_void_p, C.c_void_p]
g._L.rdet3d_debug_marks(g._h, m.ctypes.data)
m = m.astype(np.int64)[:64]
t0 = m[:, 0].min()
print("  wg  start  roots   rank  gather  sort+sum  end(us)  stretch size")
for b in range(0, 64, 3):
    r = m[b]
    print("%4d %6.2f %6.2f %6.2f %6.2f %6.2f %8.2f %8d %4d" % (b, (r[0] - t0) / 100, (r[1] - r[0]) / 100, (r[2] - r[1]) / 100, (r[3] - r[2]) / 100,
                                                         (r[4] - r[3]) / 100, (r[4] - t0) / 100, r[6], r[7]))
if len(sys.argv) > 2 and sys.argv[2] == "link":        # k3_cc_link: first query of each workgroup (marks overwritten by it)
    m = np.zeros((2048, 8), np.uint64)
    g._L.rdet3d_debug_marks(g._h, m.ctypes.data)
    m = m.astype(np.int64)
    live = m[:, 3] > 0
    live[:64] = False                                   # (k3_clusters has overwritten those)
    t0 = m[live, 0].min()
    rows = [((r[0] - t0) / 100, (r[1] - r[0]) / 100, (r[2] - r[1]) / 100, (r[3] - r[2]) / 100, (r[3] - t0) / 100, r[6], (r[5] - r[4]) / 100 if r[5] > r[4] else 0, (r[3] - r[5]) / 100 if r[5] > 0 else 0, r[7]) for r in m[live]]
    rows.sort(key=lambda x: -x[4])
    print("k3_cc_link: start  chase  boxes  sweep  end(us)  tiles  last batch: tops(us) unions+rest(us) hops*10000+rounds*100+distinct  (%d workgroups)" % len(rows))
    print('starts: median %.2f p90 %.2f max %.2f' % tuple(np.percentile([r[0] for r in rows], [50, 90, 100])))
    for r in rows[:10] + rows[-3:]:
        print("          %6.2f %6.2f %6.2f %6.2f %7.2f %5d %6.2f %6.2f %7d" % r)
