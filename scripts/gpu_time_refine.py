"""Per-iteration cost of kg_refine: RefineMatch call time against the iteration cap.  GPU box: python scripts/gpu_time_refine.py"""
import json, math, sys, time
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd.grid import GridFrontEnd, CeresScanMatcherOptions2D
from tests.grid_cases import room_grid, scan_of

g = GridFrontEnd(max_points=16384, max_cells=1024 * 1024, max_candidates=1 << 18)
cells, max_xy, occ = room_grid()
g.SetGrid(cells, 0.05, max_xy)
true = np.array([0.5, 0.3, 0.2])
for n in (64, 533, 2439, 8000):
    pts = scan_of(occ, true, n_points=n, seed=7).astype(np.float32)
    start = true + [0.01, -0.012, 0.004]
    row = {"points": n}
    for cap in (0, 1, 2, 4, 8, 100):
        o = CeresScanMatcherOptions2D(max_num_iterations=cap)
        r = g.RefineMatch(start[:2], start, pts, o)
        for _ in range(3):
            g.RefineMatch(start[:2], start, pts, o)
        t0 = time.perf_counter()
        for _ in range(50):
            g.RefineMatch(start[:2], start, pts, o)
        row[f"cap{cap}_us"] = round((time.perf_counter() - t0) / 50 * 1e6, 1)
        row[f"cap{cap}_iters"] = r.iterations
    print(json.dumps(row))
