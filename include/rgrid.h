/* rgrid.h -- C ABI of the MI355X-native grid-mapper front-end: the step right after
 * the hot path, which consumes the detector's de-skewed returns (rdet2d_get_range_data) and the EKF pose.
 *
 * It replaces, in the reference (ShihanWang/reflector_ekf_slam):
 *   sensor::VoxelFilter::Filter                 src/sensor/voxel_filter.cc:81-95   (called from map_builder.cc:30-31)
 *   sensor::AdaptiveVoxelFilter::Filter         src/sensor/voxel_filter.cc:116-120 (map_builder.cc:73)
 *   scan_matching::RealTimeCorrelativeScanMatcher2D::Match
 *                                               src/scan_matching/real_time_correlative_scan_matcher_2d.cc:84-118
 *                                               (map_builder.cc:43) with its helpers SearchParameters,
 *                                               GenerateRotatedScans, DiscretizeScans
 *                                               (correlative_scan_matcher_2d.cc:10-123)
 *   mapping::ProbabilityGridRangeDataInserter2D::Insert
 *                                               src/mapping/probability_grid_range_data_inserter_2d.cc:40-114
 *                                               (map_builder.cc: range_data_inserter_->Insert); its GrowAsNeeded /
 *                                               Grid2D::GrowLimits step (:20-38, src/mapping/grid_2d.cc:59-99) is
 *                                               rgrid_grow_as_needed
 *   scan_matching::CeresScanMatcher2D::Match    src/scan_matching/ceres_scan_matcher_2d.cc:26-62 (map_builder.cc:49-53) with
 *                                               occupied_space_cost_function_2d.cc:25-52 and the translation / rotation
 *                                               delta functors -- the Ceres solve restated (Ceres is not a pinned
 *                                               dependency of the reference: parity with a Ceres build is unpinned)
 *   mapping::ProbabilityGrid::DrawToSubmapTexture
 *                                               src/mapping/probability_grid.cc:86-131 (Submap2D::GetMapTextureData,
 *                                               MapBuilder::ToSubmapTexture, map_builder.cc:128-134) without the gzip
 *                                               container
 * Not covered: Submap2D::Finish / ComputeCroppedGrid (never called by the reference's node), IO.
 *
 * Conventions as in rekf.h / rdet.h: opaque handles, plain pointers and sizes, 0 / negative error codes,
 * caller owns every buffer, a handle is not thread-safe, calls synchronise before returning.
 */
#ifndef RGRID_H_
#define RGRID_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGRID_ABI_VERSION 3

enum {
    RGRID_OK = 0,
    RGRID_ERR_INVALID = -1,       /* bad argument */
    RGRID_ERR_HIP = -2,           /* a HIP runtime call failed */
    RGRID_ERR_CAPACITY = -4,      /* more points / cells / candidates than the handle was created for */
    RGRID_ERR_BUFFER = -5,        /* caller buffer too small */
    RGRID_ERR_EMPTY = -6          /* empty point cloud (the reference CHECK-fails, real_time_...cc:33) */
};

typedef struct rgrid rgrid_t;

/* scan_matching::RealTimeCorrelativeScanMatcherOptions (real_time_correlative_scan_matcher_2d.h:34-40);
 * defaults in the reference's caller: 0.2 m, 15 deg (in radians here), 1e-1, 1e-1 (src/ros_node.cc:329-344). */
typedef struct rgrid_match_options {
    double linear_search_window;
    double angular_search_window;
    double translation_delta_cost_weight;
    double rotation_delta_cost_weight;
} rgrid_match_options;

/* One handle = one HIP stream + device buffers for up to max_points points, max_cells grid cells and
 * max_candidates search candidates (num_scans * (2 * num_linear + 1)^2). */
int rgrid_create(int max_points, int max_cells, int max_candidates, int device, rgrid_t **out);
void rgrid_destroy(rgrid_t *h);

/* sensor::VoxelFilter(resolution).Filter(points): the first point that falls into every voxel, input
 * order kept (voxel_filter.cc:81-95).  out_xy holds out_cap points; *m = points written. */
int rgrid_voxel_filter(rgrid_t *h, const float *xy, int n, float resolution, float *out_xy, int out_cap, int *m);

/* sensor::AdaptiveVoxelFilter(options).Filter(points) (voxel_filter.cc:15-76,116-120): range gate, then the
 * coarsest voxel size <= max_length (bisection to 10 %) that still leaves min_num_points points.
 * Reference defaults: 0.9, 500, 100 (src/ros_node.cc:312-322). */
int rgrid_adaptive_voxel_filter(rgrid_t *h, const float *xy, int n, double max_length, double min_num_points,
                                double max_range, float *out_xy, int out_cap, int *m);

/* The probability grid the matcher scores against (mapping::ProbabilityGrid / Grid2D): correspondence-cost cell
 * values as the reference stores them (uint16, 0 = unknown, grid_2d.h:83-91), flat index num_x_cells * y + x
 * (grid_2d.h:102-106), MapLimits (resolution, max.x, max.y) (map_limits.h:24-45).  The cells are copied to the
 * device and stay resident until the next call. */
int rgrid_set_grid(rgrid_t *h, const uint16_t *cells, int num_x_cells, int num_y_cells, double resolution,
                   double max_x, double max_y);

/* ProbabilityGridRangeDataInserter2D::Insert (src/mapping/probability_grid_range_data_inserter_2d.cc:40-114) on the
 * resident grid: the cell of every return gets the hit table, every cell on the rays origin -> return and
 * origin -> miss the miss table (RayToPixelMask, ray_to_pixel_mask.cc:17-168, sub-pixel scale 1000), a cell is
 * updated at most once per insertion and hits win (probability_grid.cc:38-53), then FinishUpdate (grid_2d.cc:20-29).
 * hit / miss probabilities: the reference's options are float (0.55 / 0.49, src/ros_node.cc:390-396).
 * GrowAsNeeded is a call of its own (rgrid_grow_as_needed, to be made first as CastRays does, :45): here a point
 * outside the grid returns RGRID_ERR_CAPACITY and leaves the grid untouched.
 * The grid must be in the finished state (no cell with the update marker 0x8000 set). */
int rgrid_insert(rgrid_t *h, const float origin_xy[2], const float *returns_xy, int n_returns, const float *misses_xy,
                 int n_misses, float hit_probability, float miss_probability, int insert_free_space);

/* GrowAsNeeded (probability_grid_range_data_inserter_2d.cc:20-38): the float bounding box of origin, returns and
 * misses, padded by 1e-6, and Grid2D::GrowLimits (src/mapping/grid_2d.cc:59-99) for its two corners -- the grid
 * doubles in both directions (old cells in the middle, new cells unknown, max += resolution * (ny / 2, nx / 2))
 * until it contains the corner.  RGRID_ERR_CAPACITY (grid untouched) if that needs more than max_cells cells;
 * non-finite coordinates are RGRID_ERR_INVALID (the reference would loop forever). */
int rgrid_grow_as_needed(rgrid_t *h, const float origin_xy[2], const float *returns_xy, int n_returns,
                         const float *misses_xy, int n_misses);

/* MapLimits of the resident grid (any pointer may be NULL). */
int rgrid_get_limits(rgrid_t *h, int *num_x_cells, int *num_y_cells, double *resolution, double *max_x, double *max_y);

/* Copy of the resident grid cells (num_x_cells * num_y_cells values, same layout as rgrid_set_grid). */
int rgrid_get_grid(rgrid_t *h, uint16_t *cells, long cap);

/* RealTimeCorrelativeScanMatcher2D::Match (real_time_correlative_scan_matcher_2d.cc:84-118):
 * initial_pose = (x, y, rotation angle); points in the tracking frame; pose_estimate = (x, y, angle) of the best
 * candidate (first maximum in the reference's candidate order); returns its score in *score.
 * best3 (nullable) = (scan_index, x_index_offset, y_index_offset); info3 (nullable) = (num_scans,
 * num_linear_perturbations, num_candidates). */
int rgrid_match(rgrid_t *h, const rgrid_match_options *opt, const double initial_pose[3], const float *points_xy,
                int n, double pose_estimate[3], double *score, int best3[3], int info3[3]);

/* ProbabilityGrid::DrawToSubmapTexture (probability_grid.cc:86-131): the resident grid cropped to the bounding box of its
 * known cells, two bytes per cell (value, alpha), x fastest -- the string the reference gzips into SubmapTexture::cells.
 * box = (offset_x, offset_y, width, height) in cells; slice_max = limits.max - resolution * (offset_y, offset_x), the
 * translation of SubmapTexture::slice_pose before local_pose^-1 is applied (:122-126).  cells needs
 * 2 * width * height bytes (at most 2 * num_x_cells * num_y_cells); RGRID_ERR_BUFFER reports the box it would need. */
int rgrid_draw_texture(rgrid_t *h, uint8_t *cells, long cap, int box[4], double slice_max[2]);

/* scan_matching::CeresScanMatcherOptions2D (ceres_scan_matcher_2d.h:16-22) + the two ceres::Solver::Options fields the
 * reference sets (src/ros_node.cc:350-377): defaults 1.0, 0.1, 0.4, 100 iterations, non-monotonic steps on. */
typedef struct rgrid_refine_options {
    double occupied_space_weight;
    double translation_weight;
    double rotation_weight;
    int max_num_iterations;
    int use_nonmonotonic_steps;
} rgrid_refine_options;

/* What the reference reads of ceres::Solver::Summary, reduced: termination 0 = CONVERGENCE, 1 = NO_CONVERGENCE
 * (iteration limit), 2 = FAILURE (five invalid steps in a row). */
typedef struct rgrid_refine_summary {
    double initial_cost;
    double final_cost;
    int iterations;
    int termination;
} rgrid_refine_summary;

/* CeresScanMatcher2D::Match (ceres_scan_matcher_2d.cc:26-62) against the resident grid: minimises over (x, y, angle)
 *   sum_i (occupied_space_weight / sqrt(n) * bicubic correspondence cost at point i)^2
 *   + (translation_weight * (xy - target_translation))^2 + (rotation_weight * (angle - initial angle))^2
 * with Ceres' Levenberg-Marquardt trust-region loop at its default tolerances.  initial_pose_estimate = the
 * correlative matcher's answer, target_translation = the prediction's translation (map_builder.cc:49-53).
 * summary may be NULL. */
int rgrid_refine_match(rgrid_t *h, const rgrid_refine_options *opt, const double target_translation[2],
                       const double initial_pose[3], const float *points_xy, int n, double pose_estimate[3],
                       rgrid_refine_summary *summary);

/* mapping::MapBuilderOptions (include/mapping/map_builder.h:22-30), flattened; defaults = src/ros_node.cc:299-396:
 * 0.05, 0.025, {0.9, 500, 100}, {0.2, 15 deg in rad, 0.1, 0.1}, {1, 0.1, 0.4, 100, 1}, 0.55, 0.49, 1. */
typedef struct rgrid_map_builder_options {
    float resolution;
    float voxel_filter_size;
    double adaptive_max_length, adaptive_min_num_points, adaptive_max_range;
    rgrid_match_options match;
    rgrid_refine_options refine;
    float hit_probability, miss_probability;
    int insert_free_space;
} rgrid_map_builder_options;

enum { RGRID_SCAN_INSERTED = 0, RGRID_SCAN_DROPPED_EMPTY = 1, RGRID_SCAN_FILTERED_EMPTY = 2 };

/* mapping::MapBuilder::AddRangeData (src/mapping/map_builder.cc:57-108) in one call, on the handle's resident grid (created
 * on the first call as MapBuilder::CreateGrid does, :112-126: 100 x 100 cells around the first origin): gravity
 * alignment + voxel filters (:20-32), adaptive filter (:72-73), ScanMatch = correlative match + refinement (:34-55),
 * InsertIntoSubmap = grow + insert (:110-120).  range_data (origin, returns, misses) is in the tracking frame, ekf_pose =
 * (x, y, yaw) as the node builds it (src/ros_node.cc:547-549).  local_pose = Project2D(MatchingResult::local_pose);
 * returns_in_local (nullable, 2 * n_returns floats) = MatchingResult::range_data_in_local.returns.  *status tells what the
 * reference would have returned: a result (RGRID_SCAN_INSERTED) or nullptr (the other two). */
int rgrid_add_range_data(rgrid_t *h, const rgrid_map_builder_options *opt, const float origin_xy[2], const float *returns_xy,
                         int n_returns, const float *misses_xy, int n_misses, const double ekf_pose[3], double local_pose[3],
                         float *returns_in_local, int *status);

const char *rgrid_strerror(int code);
const char *rgrid_last_hip_error(rgrid_t *h);
int rgrid_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RGRID_H_ */
