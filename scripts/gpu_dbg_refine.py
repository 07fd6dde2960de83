"""Cycle breakdown of kg_refine from a -DRGRID_DEBUG_TIMING build of librgrid.so (thread 0's clock, summed over the
iterations): make -C reflector_ekf_slam_amd/csrc HIPFLAGS+=... first.  GPU box: python scripts/gpu_dbg_refine.py"""
import ctypes as C, json, sys
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd.grid import GridFrontEnd, CeresScanMatcherOptions2D
from tests.grid_cases import room_grid, scan_of

g = GridFrontEnd(max_points=16384, max_cells=1024 * 1024, max_candidates=1 << 18)
cells, max_xy, occ = room_grid()
g.SetGrid(cells, 0.05, max_xy)
true = np.array([0.5, 0.3, 0.2])
names = ["head", "eval+wave_reduce", "barrier1", "totals", "judge", "next_candidate", "barrier2", "-"]
for n in (64, 533, 2439):
    pts = scan_of(occ, true, n_points=n, seed=7).astype(np.float32)
    start = true + [0.01, -0.012, 0.004]
    for _ in range(3):
        r = g.RefineMatch(start[:2], start, pts)
    out = (C.c_longlong * 16)()
    g._L.rgrid_debug_refine_cycles.argtypes = [C.c_void_p, C.c_void_p]
    g._L.rgrid_debug_refine_cycles(g._h, out)
    it = max(r.iterations, 1)
    print(json.dumps({"points": n, "iterations": r.iterations, **{names[k]: round(out[k] / it) for k in range(7)},
                      "eval": {n_: round(out[8 + k] / it) for k, n_ in enumerate(["sincos", "loads+row_splines", "column_splines+sums", "loop_exit", "wave_reduce"])}, "unit": "clock64 ticks per iteration"}))
