"""Per-scan latency of the MapBuilder mirror (reflector_ekf_slam_amd/map_builder.py: gravity alignment, 2 voxel filters,
adaptive filter, correlative match, LM refinement, growth + insertion) on the GPU against the same host logic over
the CPU oracle.  GPU box: python scripts/gpu_bench_map_builder.py [scans]"""
import json, math, sys, time
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd.map_builder import MapBuilder, MapBuilderOptions, RangeData
from tests.grid_cases import room_grid, scan_of
from tests.oracle_front_end import OracleFrontEnd

nscans = int(sys.argv[1]) if len(sys.argv) > 1 else 40
_, _, occ = room_grid()
t = np.linspace(0.0, 1.0, nscans)
poses = np.stack([0.4 + 2.5 * t, -0.3 + 1.2 * np.sin(2.0 * t), 0.2 + 1.4 * t], 1)
rng = np.random.default_rng(17)
scans = []
for k, true in enumerate(poses):
    pts = scan_of(occ, true, n_points=3600, seed=300 + k).astype(np.float32)
    ang = rng.uniform(-math.pi, math.pi, 100)
    misses = np.stack([6.0 * np.cos(ang), 6.0 * np.sin(ang)], 1).astype(np.float32)
    scans.append((RangeData(np.zeros(2, np.float32), pts, misses), true + rng.normal(0, 1, 3) * [0.02, 0.02, 0.005]))
for name, mb in (("gpu", MapBuilder(MapBuilderOptions(), max_points=16384, max_cells=2048 * 2048)), ("cpu_oracle", MapBuilder(MapBuilderOptions(), front_end=OracleFrontEnd()))):
    times = []
    n = nscans if name == "gpu" else min(nscans, 8)
    for k in range(n):
        t0 = time.perf_counter()
        r = mb.AddRangeData(float(k), scans[k][0], scans[k][1])
        times.append(time.perf_counter() - t0)
    (cells, lim) = mb.grid()
    print(json.dumps({"backend": name, "scans": n, "returns_per_scan": 3600, "median_ms_per_scan": round(1e3 * float(np.median(times[1:])), 3),
                      "grid": [lim[0], lim[1]], "iterations_last": mb.last_summary.iterations, "pose_err_last": [float(v) for v in np.abs(r.local_pose - poses[n - 1])]}))

# the same pipeline as ONE C call per scan (rgrid_add_range_data)
from reflector_ekf_slam_amd.grid import GridFrontEnd
fe = GridFrontEnd(max_points=16384, max_cells=2048 * 2048)
opt = MapBuilderOptions()
times = []
for k in range(nscans):
    t0 = time.perf_counter()
    st, pose, _ = fe.AddRangeData(opt, scans[k][0].origin, scans[k][0].returns, scans[k][0].misses, scans[k][1])
    times.append(time.perf_counter() - t0)
lim = fe.GetLimits()
print(json.dumps({"backend": "gpu_one_c_call", "scans": nscans, "returns_per_scan": 3600, "median_ms_per_scan": round(1e3 * float(np.median(times[1:])), 3),
                  "grid": [lim[0], lim[1]], "pose_err_last": [float(v) for v in np.abs(pose - poses[nscans - 1])]}))
