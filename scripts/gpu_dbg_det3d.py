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
m_all = np.zeros((6, 2048, 8), np.uint64)
g._L.rdet3d_debug_marks.argtypes = [C.c_void_p, C.c_void_p]
g._L.rdet3d_debug_marks(g._h, m_all.ctypes.data)
m_all = m_all.astype(np.int64)
r = m_all[4][int(np.argmax(m_all[4][:, 0]))]            # (k3f_front: the tile workgroup that arrived last)
print("k3f_sort (one workgroup): prefix %.2f  survivors in registers %.2f  histogram %.2f  scan %.2f  sorted in LDS %.2f  out + boxes %.2f  = %.2f us" % (
    (r[1] - r[0]) / 100, (r[2] - r[1]) / 100, (r[3] - r[2]) / 100, (r[4] - r[3]) / 100, (r[5] - r[4]) / 100, (r[6] - r[5]) / 100, (r[6] - r[0]) / 100))
for b in range(0, 8):
    r = m_all[5][b]
    if r[6] > r[0] > 0:
        print("k3f_clusters wg %d (wave 0): tables in LDS %.2f  roots + sizes %.2f  gate %.2f  rank %.2f  (grid) members %.2f  sum + publish %.2f  = %.2f us" % (
            b, (r[1] - r[0]) / 100, (r[2] - r[1]) / 100, (r[3] - r[2]) / 100, (r[4] - r[3]) / 100, (r[5] - r[4]) / 100, (r[6] - r[5]) / 100, (r[6] - r[0]) / 100))
m = m_all[0][:64]
t0 = m[:, 0].min()
print("k3_clusters")
print("  wg  start  roots   rank  gather  sort+sum  end(us)  stretch size")
for b in range(0, 64, 3):
    r = m[b]
    print("%4d %6.2f %6.2f %6.2f %6.2f %6.2f %8.2f %8d %4d" % (b, (r[0] - t0) / 100, (r[1] - r[0]) / 100, (r[2] - r[1]) / 100, (r[3] - r[2]) / 100,
                                                         (r[4] - r[3]) / 100, (r[4] - t0) / 100, r[6], r[7]))


def pct(v):
    return "med %.2f p90 %.2f max %.2f" % tuple(np.percentile(v, [50, 90, 100]))


# k3_knn (wave 0 of every workgroup): entry, first loads back, first sort done, sweep done, end; steps (pairs of tiles) and merges
k = m_all[2]
t0 = k[k[:, 0] > 0, 0].min()
live = (k[:, 4] > k[:, 0]) & (k[:, 0] >= t0) & (k[:, 1] > k[:, 0])
k = k[live]
print("k3_knn: %d workgroups with a query; kernel spans %.2f us" % (len(k), (k[:, 4].max() - t0) / 100))
print("  entry        " + pct((k[:, 0] - t0) / 100))
print("  loads back   " + pct((k[:, 1] - k[:, 0]) / 100))
print("  first sort   " + pct((k[:, 2] - k[:, 1]) / 100))
print("  sweep        " + pct((k[:, 3] - k[:, 2]) / 100))
print("  sum + store  " + pct((k[:, 4] - k[:, 3]) / 100))
print("  end          " + pct((k[:, 4] - t0) / 100))
print("  steps        " + pct(k[:, 6]) + "   merges " + pct(k[:, 7]))
order = np.argsort(-(k[:, 4] - t0))
for r in k[order[:6]]:
    print("     slowest: entry %.2f loads %.2f sort %.2f sweep %.2f tail %.2f end %.2f steps %d merges %d" % (
        (r[0] - t0) / 100, (r[1] - r[0]) / 100, (r[2] - r[1]) / 100, (r[3] - r[2]) / 100, (r[4] - r[3]) / 100, (r[4] - t0) / 100, r[6], r[7]))
# k3_cc_min: entry, M known, threshold known, first query done; tiles opened
k = m_all[3]
t0 = k[k[:, 0] > 0, 0].min()
live = (k[:, 3] > k[:, 0]) & (k[:, 2] > k[:, 0])
k = k[live]
print("k3_cc_min: %d workgroups; kernel spans %.2f us" % (len(k), (k[:, 3].max() - t0) / 100))
print("  entry        " + pct((k[:, 0] - t0) / 100))
print("  M known      " + pct((k[:, 1] - k[:, 0]) / 100))
print("  threshold    " + pct((k[:, 2] - k[:, 1]) / 100))
print("  sweep+store  " + pct((k[:, 3] - k[:, 2]) / 100))
print("  end          " + pct((k[:, 3] - t0) / 100))
print("  tiles        " + pct(k[:, 6]))
# k3_cc_link: first query of each workgroup: own chain, boxes, all batches (loads / chases of the neighbours' tops / unions), end
k = m_all[1]
t0 = k[k[:, 0] > 0, 0].min()
live = (k[:, 3] > k[:, 0]) & (k[:, 0] >= t0) & (k[:, 3] - k[:, 0] < 10000)
k = k[live]
print("k3_cc_link: %d workgroups whose first query is an inlier; kernel spans %.2f us" % (len(k), (k[:, 3].max() - t0) / 100))
print("  entry        " + pct((k[:, 0] - t0) / 100))
print("  own chain    " + pct((k[:, 1] - k[:, 0]) / 100))
print("  boxes        " + pct((k[:, 2] - k[:, 1]) / 100))
print("  batches      " + pct((k[:, 3] - k[:, 2]) / 100))
print("    loads      " + pct(k[:, 4] / 100))
print("    chases     " + pct(k[:, 5] / 100))
print("    unions     " + pct(k[:, 7] / 100))
print("  end          " + pct((k[:, 3] - t0) / 100))
lo = k[:, 6] & 0xffffffff
print("  tiles " + pct(k[:, 6] >> 32) + "  batches " + pct(lo // 10000) + "  hops " + pct((lo // 100) % 100) + "  union rounds " + pct(lo % 100))
order = np.argsort(-(k[:, 3] - t0))
for r in k[order[:8]]:
    l = r[6] & 0xffffffff
    print("     slowest: entry %.2f chain %.2f boxes %.2f batches %.2f (loads %.2f chases %.2f unions %.2f) end %.2f  tiles %d batches %d hops %d rounds %d" % (
        (r[0] - t0) / 100, (r[1] - r[0]) / 100, (r[2] - r[1]) / 100, (r[3] - r[2]) / 100, r[4] / 100, r[5] / 100, r[7] / 100, (r[3] - t0) / 100,
        r[6] >> 32, l // 10000, (l // 100) % 100, l % 100))
