"""The CPU oracle behind GridFrontEnd's interface: lets tests run reflector_ekf_slam_amd.map_builder.MapBuilder's host
logic over the oracle's restatements and compare the whole pipeline with the GPU one.  TEST INFRASTRUCTURE."""
from __future__ import annotations

import numpy as np

from oracle import binding as ob
from reflector_ekf_slam_amd.grid import MatchResult, RefineResult


class OracleFrontEnd:
    def __init__(self):
        self.cells = None
        self.res = None
        self.max_xy = None

    def VoxelFilter(self, pts, size):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        return ob.oracle_voxel_filter(pts, size) if pts.shape[0] else pts

    def AdaptiveVoxelFilter(self, pts, o):
        return ob.oracle_adaptive_voxel_filter(np.ascontiguousarray(pts, np.float32).reshape(-1, 2), o.max_length, o.min_num_points, o.max_range)

    def SetGrid(self, cells, res, max_xy):
        self.cells, self.res, self.max_xy = np.array(cells, np.uint16), float(res), (float(max_xy[0]), float(max_xy[1]))

    def Match(self, pose, pts, o):
        score, est, best, info = ob.oracle_match(pose, pts, self.cells, self.res, self.max_xy, o.linear_search_window, o.angular_search_window,
                                                 o.translation_delta_cost_weight, o.rotation_delta_cost_weight)
        return MatchResult(score, np.array(est), tuple(best), tuple(info))

    def RefineMatch(self, target, init, pts, o):
        pose, s = ob.oracle_refine_match(target, init, pts, self.cells, self.res, self.max_xy, o.occupied_space_weight, o.translation_weight,
                                         o.rotation_weight, o.max_num_iterations, o.use_nonmonotonic_steps)
        return RefineResult(pose, s["initial_cost"], s["final_cost"], s["iterations"], s["termination"])

    def Insert(self, origin, returns, misses, o, grow=True):
        if grow:
            self.cells, self.max_xy, _ = ob.oracle_grow(self.cells, self.res, self.max_xy, origin, returns, misses)
        self.cells = ob.oracle_insert(self.cells, self.res, self.max_xy, origin, returns, misses, o.hit_probability, o.miss_probability,
                                      o.insert_free_space)

    def GetGrid(self):
        return self.cells.copy()

    def GetLimits(self):
        return self.cells.shape[1], self.cells.shape[0], self.res, self.max_xy[0], self.max_xy[1]

    def DrawTexture(self):
        return ob.oracle_draw_texture(self.cells, self.res, self.max_xy)
