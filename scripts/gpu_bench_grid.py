"""Latency of the grid front-end through the C ABI (host buffers in, result out, synchronising calls) next to the
CPU oracle on the same inputs.  Usage (GPU box): python scripts/gpu_bench_grid.py [reps] -> one JSON line per case."""
import json, math, sys, time
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd.grid import GridFrontEnd, AdaptiveVoxelFilterOptions
from oracle.binding import oracle_voxel_filter, oracle_adaptive_voxel_filter, oracle_match, oracle_insert, oracle_refine_match
from tests.grid_cases import room_grid, scan_of

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def timeit(f, n):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e6


g = GridFrontEnd(max_points=16384, max_cells=1024 * 1024, max_candidates=1 << 18)
rng = np.random.default_rng(3)
cells, max_xy, occ = room_grid()
g.SetGrid(cells, 0.05, max_xy)
returns = scan_of(occ, (0.5, 0.3, 0.2), n_points=3600, noise=0.01)
vf = g.VoxelFilter(returns, 0.025)
print(json.dumps({"case": "voxel_filter_0.025", "points": int(returns.shape[0]), "kept": int(vf.shape[0]),
                  "identical_to_oracle": bool(np.array_equal(vf, oracle_voxel_filter(returns, 0.025))),
                  "gpu_call_us": round(timeit(lambda: g.VoxelFilter(returns, 0.025), reps), 1),
                  "cpu_oracle_us": round(timeit(lambda: oracle_voxel_filter(returns, 0.025), max(reps // 10, 3)), 1)}))
opt = AdaptiveVoxelFilterOptions()
av = g.AdaptiveVoxelFilter(vf, opt)
print(json.dumps({"case": "adaptive_voxel_filter", "points": int(vf.shape[0]), "kept": int(av.shape[0]),
                  "identical_to_oracle": bool(np.array_equal(av, oracle_adaptive_voxel_filter(vf, opt.max_length, opt.min_num_points, opt.max_range))),
                  "gpu_call_us": round(timeit(lambda: g.AdaptiveVoxelFilter(vf, opt), reps), 1),
                  "cpu_oracle_us": round(timeit(lambda: oracle_adaptive_voxel_filter(vf, opt.max_length, opt.min_num_points, opt.max_range), max(reps // 10, 3)), 1)}))
true = np.array([0.5, 0.3, 0.2])
init = true + [0.08, -0.12, math.radians(5.0)]
for name, pts in (("match_adaptive_cloud", av), ("match_full_cloud", vf)):
    r = g.Match(init, pts)
    sc, pose, best, info = oracle_match(init, pts, cells, 0.05, max_xy)
    print(json.dumps({"case": name, "points": int(pts.shape[0]), "num_scans": info[0], "candidates": info[2],
                      "same_candidate": bool(r.best == best), "score": r.score, "score_rel_diff": abs(r.score - sc) / sc,
                      "gpu_call_us": round(timeit(lambda: g.Match(init, pts), reps), 1),
                      "cpu_oracle_us": round(timeit(lambda: oracle_match(init, pts, cells, 0.05, max_xy), 3), 1)}))
    if name == "match_adaptive_cloud":                      # the refinement MapBuilder::ScanMatch runs next (map_builder.cc:49-53)
        rr = g.RefineMatch(init[:2], r.pose_estimate, pts)
        po, so = oracle_refine_match(init[:2], r.pose_estimate, pts, cells, 0.05, max_xy)
        print(json.dumps({"case": "refine_match_adaptive_cloud", "points": int(pts.shape[0]), "iterations": rr.iterations,
                          "oracle_iterations": so["iterations"], "pose_abs_diff_vs_oracle": float(np.abs(rr.pose_estimate - po).max()),
                          "pose_error_m_rad": [float(v) for v in np.abs(rr.pose_estimate - true)],
                          "gpu_call_us": round(timeit(lambda: g.RefineMatch(init[:2], r.pose_estimate, pts), reps), 1),
                          "cpu_oracle_us": round(timeit(lambda: oracle_refine_match(init[:2], r.pose_estimate, pts, cells, 0.05, max_xy), 10), 1)}))

# range-data insertion into an initially unknown grid (3600 returns + 100 misses), then cell-for-cell comparison
pose = (0.5, 0.3, 0.2)
c, s_ = math.cos(pose[2]), math.sin(pose[2])
world = np.stack([pose[0] + c * returns[:, 0] - s_ * returns[:, 1], pose[1] + s_ * returns[:, 0] + c * returns[:, 1]], 1).astype(np.float32)
ang = rng.uniform(-math.pi, math.pi, 100)
misses = np.stack([pose[0] + 5.0 * np.cos(ang), pose[1] + 3.5 * np.sin(ang)], 1).astype(np.float32)
origin = np.array(pose[:2], np.float32)
empty = np.zeros_like(cells)
g.SetGrid(empty, 0.05, max_xy)
g.Insert(origin, world, misses)
same = bool(np.array_equal(g.GetGrid(), oracle_insert(empty, 0.05, max_xy, origin, world, misses)))
print(json.dumps({"case": "insert_range_data", "returns": int(world.shape[0]), "misses": 100, "cells": int(empty.size),
                  "identical_to_oracle": same,
                  "gpu_call_us": round(timeit(lambda: g.Insert(origin, world, misses), reps), 1),
                  "cpu_oracle_us": round(timeit(lambda: oracle_insert(empty, 0.05, max_xy, origin, world, misses), 5), 1)}))
