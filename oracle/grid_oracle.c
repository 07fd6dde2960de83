/*
 * oracle/grid_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the grid-mapper front-end that consumes the detector's de-skewed returns and the
 * EKF pose:
 *
 *   sensor::VoxelFilter::Filter, sensor::AdaptiveVoxelFilter::Filter     src/sensor/voxel_filter.cc:12-118
 *   scan_matching::SearchParameters, GenerateRotatedScans, DiscretizeScans
 *                                                   src/scan_matching/correlative_scan_matcher_2d.cc:10-123
 *   scan_matching::RealTimeCorrelativeScanMatcher2D::Match / ScoreCandidates
 *                                       src/scan_matching/real_time_correlative_scan_matcher_2d.cc:20-136
 *   mapping::MapLimits::GetCellIndex / Contains                          include/mapping/map_limits.h:47-72
 *   mapping::ProbabilityGrid::GetProbability, value tables               src/mapping/probability_grid.cc:56-62,
 *                                  src/mapping/probability_values.cc:11-20, include/mapping/probability_values.h:53-57
 *   mapping::ProbabilityGridRangeDataInserter2D::Insert, RayToPixelMask, GrowAsNeeded / Grid2D::GrowLimits
 *                 src/mapping/probability_grid_range_data_inserter_2d.cc:20-114, ray_to_pixel_mask.cc:17-168, grid_2d.cc:59-99
 *   scan_matching::CeresScanMatcher2D::Match (Ceres restated)           src/scan_matching/ceres_scan_matcher_2d.cc:26-62
 *   mapping::ProbabilityGrid::DrawToSubmapTexture                        src/mapping/probability_grid.cc:86-131
 *
 * The float32 point algebra of the reference goes through Eigen (Quaternionf from AngleAxisf, quaternion *
 * UnitX, Rotation2Df * Vector2f, Translation2f) and transform::GetYaw (transform.h:27-33); Eigen is not in
 * this image, so those expressions are restated from Eigen 3.3's formulas, operation by
 * operation in float32: PARITY UNPINNED.  Compile with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- sensor::VoxelFilter::Filter (voxel_filter.cc:81-95): keep the FIRST point of every voxel, in order.
 * cell = RoundToInt(point / resolution) per axis (float division, lround: :105-110; port.h:25). */
static int voxel_key(float v, float res) { return (int)lroundf(v / res); }

int ogrid_voxel_filter(const float *xy, int n, float res, float *out_xy)
{
    int m = 0;
    int *kx = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int *ky = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        const int cx = voxel_key(xy[2 * i], res), cy = voxel_key(xy[2 * i + 1], res);
        int seen = 0;
        for (int j = 0; j < m && !seen; ++j) seen = (kx[j] == cx && ky[j] == cy);     /* unordered_set::insert(...).second */
        if (!seen) { kx[m] = cx; ky[m] = cy; out_xy[2 * m] = xy[2 * i]; out_xy[2 * m + 1] = xy[2 * i + 1]; ++m; }
    }
    free(kx); free(ky);
    return m;
}

/* ---- sensor::AdaptiveVoxelFilter::Filter (voxel_filter.cc:15-27, :29-76, :116-120) */
int ogrid_adaptive_voxel_filter(const float *xy, int n, double max_length, double min_num_points, double max_range,
                                float *out_xy)
{
    float *in = (float *)malloc(sizeof(float) * 2 * (size_t)(n > 0 ? n : 1));
    float *cand = (float *)malloc(sizeof(float) * 2 * (size_t)(n > 0 ? n : 1));
    int ni = 0;
    const float mr = (float)max_range;                                   /* FilterByMaxRange(const float max_range) */
    for (int i = 0; i < n; ++i) {
        const float x = xy[2 * i], y = xy[2 * i + 1];
        if (sqrtf(x * x + y * y) <= mr) { in[2 * ni] = x; in[2 * ni + 1] = y; ++ni; }      /* point.norm() <= max_range */
    }
    int m;
    if ((double)ni <= min_num_points) {                                  /* :33-37 already sparse enough */
        memcpy(out_xy, in, sizeof(float) * 2 * (size_t)ni);
        m = ni;
        goto done;
    }
    m = ogrid_voxel_filter(in, ni, (float)max_length, out_xy);           /* :38 VoxelFilter(float size) */
    if ((double)m >= min_num_points) goto done;                          /* :39-43 */
    for (float high = (float)max_length; high > 1e-2f * (float)max_length; high /= 2.f) {   /* :47-48 */
        float low = high / 2.f;
        m = ogrid_voxel_filter(in, ni, low, out_xy);
        if ((double)m >= min_num_points) {
            while ((high - low) / low > 1e-1f) {                         /* :57 */
                const float mid = (low + high) / 2.f;
                const int mc = ogrid_voxel_filter(in, ni, mid, cand);
                if ((double)mc >= min_num_points) { low = mid; m = mc; memcpy(out_xy, cand, sizeof(float) * 2 * (size_t)mc); }
                else high = mid;
            }
            goto done;
        }
    }
done:
    free(in); free(cand);
    return m;
}

/* ---- float32 helpers restating Eigen ------------------------------------------------------------
 * Project2D(Rigid3f::Rotation(AngleAxisf(angle, UnitZ))) (correlative_scan_matcher_2d.cc:95-98,
 * real_time_correlative_scan_matcher_2d.cc:92-97): Quaternionf(AngleAxisf): w = cos(angle/2), z = sin(angle/2);
 * GetYaw = atan2 of (q * UnitX) with Eigen's  v + w*uv + vec x uv,  uv = 2 (vec x v);  the Rigid2f then rotates
 * with Rotation2Df(yaw): (c x - s y, s x + c y). */
void ogrid_rotation_cs(float angle, float *c, float *s)
{
    const float ha = 0.5f * angle;
    const float w = cosf(ha), z = sinf(ha);
    const float uvy = z + z;                       /* uv = vec x UnitX = (0, z, 0); uv += uv */
    const float dx = (1.f + w * 0.f) + (0.f * 0.f - z * uvy);
    const float dy = (0.f + w * uvy) + (z * 0.f - 0.f * 0.f);
    const float yaw = atan2f(dy, dx);
    *c = cosf(yaw); *s = sinf(yaw);
}
static void rotate_cloud(const float *in, int n, float c, float s, float *out)
{
    for (int i = 0; i < n; ++i) {
        const float x = in[2 * i], y = in[2 * i + 1];
        out[2 * i] = c * x - s * y;
        out[2 * i + 1] = s * x + c * y;
    }
}

/* ---- probability of a cell value (probability_values.cc:11-20, probability_values.h:53-57,
 * probability_grid.cc:56-62): out of the grid -> kMinProbability; value 0 (unknown) -> 1 - kMaxCorrespondenceCost */
static float og_value_to_cost(uint16_t value)                  /* ValueToCorrespondenceCost (probability_values.h:53-57) */
{
    const float kMinProbability = 0.1f, kMaxProbability = 1.f - kMinProbability;
    const float lower = 1.f - kMaxProbability /* kMinCorrespondenceCost */, upper = 1.f - kMinProbability /* kMax... */;
    const uint16_t v = (uint16_t)(value & 32767u);             /* the table repeats for values with the update marker */
    if (v == 0) return upper;                                   /* unknown_result = kMaxCorrespondenceCost */
    const float kScale = (upper - lower) / (32768 - 2.f);
    return v * kScale + (lower - kScale);
}
float ogrid_value_to_probability(uint16_t value)
{
    return 1.f - og_value_to_cost(value);                       /* CorrespondenceCostToProbability */
}

typedef struct {
    double linear_search_window, angular_search_window, translation_delta_cost_weight, rotation_delta_cost_weight;
} ogrid_match_options;

/* ---- RealTimeCorrelativeScanMatcher2D::Match (real_time_correlative_scan_matcher_2d.cc:84-118).
 * grid: cells[num_x_cells * y + x] (grid_2d.h:102-106), limits (resolution, max_x, max_y).
 * out: pose_estimate[3] (x, y, angle = initial + best orientation), best[3] = (scan_index, x_off, y_off),
 * returns the best score.  info[0..2] = num_scans, num_linear_perturbations, num_candidates. */
double ogrid_match(const ogrid_match_options *opt, const double initial_pose[3], const float *points_xy, int n,
                   const uint16_t *cells, int num_x_cells, int num_y_cells, double resolution, double max_x, double max_y,
                   double pose_estimate[3], int best[3], int info[3])
{
    float *rot0 = (float *)malloc(sizeof(float) * 2 * (size_t)(n > 0 ? n : 1));
    float *rot = (float *)malloc(sizeof(float) * 2 * (size_t)(n > 0 ? n : 1));
    int *dix = (int *)malloc(sizeof(int) * 2 * (size_t)(n > 0 ? n : 1));
    float c0, s0;
    ogrid_rotation_cs((float)initial_pose[2], &c0, &s0);        /* initial_rotation.cast<float>().angle() */
    rotate_cloud(points_xy, n, c0, s0, rot0);
    /* SearchParameters (correlative_scan_matcher_2d.cc:10-42) */
    float max_scan_range = 3.f * (float)resolution;
    for (int i = 0; i < n; ++i) {
        const float x = rot0[2 * i], y = rot0[2 * i + 1];
        const float range = sqrtf(x * x + y * y);
        if (range > max_scan_range) max_scan_range = range;      /* std::max(range, max_scan_range) */
    }
    const double kSafetyMargin = 1. - 1e-3;
    const double step = kSafetyMargin * acos(1. - (resolution * resolution) / (2. * (double)(max_scan_range * max_scan_range)));
    const int num_angular = (int)ceil(opt->angular_search_window / step);
    const int num_scans = 2 * num_angular + 1;
    const int num_linear = (int)ceil(opt->linear_search_window / resolution);
    const float tx = (float)initial_pose[0], ty = (float)initial_pose[1];       /* Eigen::Translation2f(x, y) */
    double best_score_d = 0;
    float best_score = 0.f;
    int have = 0, ncand = 0;
    double delta_theta = -num_angular * step;
    for (int scan = 0; scan < num_scans; ++scan, delta_theta += step) {          /* GenerateRotatedScans :86-101 */
        float c, s;
        ogrid_rotation_cs((float)delta_theta, &c, &s);
        rotate_cloud(rot0, n, c, s, rot);
        for (int i = 0; i < n; ++i) {                                            /* DiscretizeScans :103-123 */
            const float px = rot[2 * i] + tx, py = rot[2 * i + 1] + ty;
            dix[2 * i] = (int)lround((max_y - (double)py) / resolution - 0.5);   /* GetCellIndex: (x index from y!) */
            dix[2 * i + 1] = (int)lround((max_x - (double)px) / resolution - 0.5);
        }
        const double orientation = (scan - num_angular) * step;
        for (int xo = -num_linear; xo <= num_linear; ++xo)
            for (int yo = -num_linear; yo <= num_linear; ++yo) {                 /* GenerateExhaustiveSearchCandidates :60-76 */
                float score = 0.f;
                for (int i = 0; i < n; ++i) {                                    /* ComputeCandidateScore :20-36 */
                    const int cx = dix[2 * i] + xo, cy = dix[2 * i + 1] + yo;
                    float p = 0.1f;                                              /* !Contains -> kMinProbability */
                    if (cx >= 0 && cy >= 0 && cx < num_x_cells && cy < num_y_cells) p = ogrid_value_to_probability(cells[num_x_cells * cy + cx]);
                    score += p;
                }
                score /= (float)n;
                const double x = -yo * resolution, y = -xo * resolution;         /* Candidate2D ctor, .h:62-66 */
                const double a = hypot(x, y) * opt->translation_delta_cost_weight + fabs(orientation) * opt->rotation_delta_cost_weight;
                score = (float)((double)score * exp(-(a * a)));                  /* :127-133, common::Pow2 = a * a */
                ++ncand;
                if (!have || score > best_score) {                               /* std::max_element: first maximum */
                    have = 1; best_score = score; best_score_d = score;
                    best[0] = scan; best[1] = xo; best[2] = yo;
                    pose_estimate[0] = initial_pose[0] + x;
                    pose_estimate[1] = initial_pose[1] + y;
                    pose_estimate[2] = initial_pose[2] + orientation;            /* Rotation2Dd product adds the angles */
                }
            }
    }
    if (info) { info[0] = num_scans; info[1] = num_linear; info[2] = ncand; }
    free(rot0); free(rot); free(dix);
    return best_score_d;
}

/* ===== range-data inserter: ProbabilityGridRangeDataInserter2D::Insert ==================================
 *   tables                ComputeLookupTableToApplyCorrespondenceCostOdds   src/mapping/probability_values.cc:76-96
 *   CastRays              src/mapping/probability_grid_range_data_inserter_2d.cc:40-92
 *   RayToPixelMask        src/mapping/ray_to_pixel_mask.cc:17-168 (restated below with one column-walk for both slopes)
 *   ApplyLookupTable      src/mapping/probability_grid.cc:38-53   (a cell is updated at most once per insertion)
 *   FinishUpdate          src/mapping/grid_2d.cc:20-29
 * GrowAsNeeded / GrowLimits (:19-38, grid_2d.cc:47-98) is a step of its own (ogrid_grow_limits / ogrid_grow_copy below):
 * here anything outside the grid is an error (return -1) and leaves the grid untouched. */
#define OG_MARKER 32768u
#define OG_SUBPIXEL 1000

/* kValueToCorrespondenceCost (probability_values.cc:11-20,46-51): og_value_to_cost above */
static uint16_t og_cost_to_value(float c)     /* CorrespondenceCostToValue -> BoundedFloatToValue (probability_values.h:15-29,68-72) */
{
    const float kMinProbability = 0.1f, kMaxProbability = 1.f - kMinProbability;
    const float lower = 1.f - kMaxProbability, upper = 1.f - kMinProbability;
    float cl = c;
    if (cl > upper) cl = upper;
    if (cl < lower) cl = lower;
    return (uint16_t)((int)lroundf((cl - lower) * (32766.f / (upper - lower))) + 1);
}
/* table[cell] for cell in [0, 32768): value after applying `odds`, with the update marker set */
void ogrid_lookup_table(float probability, uint16_t *table)
{
    const float odds = probability / (1.f - probability);                       /* Odds() */
    {
        const float p = odds / (odds + 1.f);                                    /* ProbabilityFromOdds */
        table[0] = (uint16_t)(og_cost_to_value(1.f - p) + OG_MARKER);
    }
    for (int cell = 1; cell != 32768; ++cell) {
        const float pc = 1.f - og_value_to_cost(cell);                          /* CorrespondenceCostToProbability */
        const float o = odds * (pc / (1.f - pc));
        const float p = o / (o + 1.f);
        table[cell] = (uint16_t)(og_cost_to_value(1.f - p) + OG_MARKER);
    }
}

typedef struct { uint16_t *cells; int nx, ny; const uint16_t *table; } og_apply_ctx;
static void og_apply(og_apply_ctx *g, int cx, int cy)                           /* ApplyLookupTable */
{
    uint16_t *cell = &g->cells[(size_t)g->nx * cy + cx];
    if (*cell >= OG_MARKER) return;
    *cell = g->table[*cell];
}

/* every pixel that contains part of the segment between the two superscaled points (ray_to_pixel_mask.cc:17-168);
 * the reference de-duplicates consecutive pixels, applying a table twice is a no-op, so no de-duplication here */
static void og_ray(og_apply_ctx *g, int bx, int by, int ex, int ey)
{
    const int S = OG_SUBPIXEL;
    if (bx > ex) { int t = bx; bx = ex; ex = t; t = by; by = ey; ey = t; }     /* ordered by x (:24-27) */
    if (bx / S == ex / S) {                                                     /* one pixel column (:35-47) */
        const int x = bx / S, y0 = (by < ey ? by : ey) / S, y1 = (by < ey ? ey : by) / S;
        for (int y = y0; y <= y1; ++y) og_apply(g, x, y);
        return;
    }
    const long long dx = ex - bx, dy = ey - by, den = 2LL * S * dx;
    int cx = bx / S, cy = by / S;
    og_apply(g, cx, cy);
    long long sub_y = (2LL * (by % S) + 1) * dx;                                /* (:64) */
    const int first_pixel = 2 * S - 2 * (bx % S) - 1, last_pixel = 2 * (ex % S) + 1, end_x = ex / S;
    sub_y += dy * first_pixel;
    const int up = dy > 0;
    for (;;) {
        og_apply(g, cx, cy);
        if (up) { while (sub_y > den) { sub_y -= den; ++cy; og_apply(g, cx, cy); } }
        else { while (sub_y < 0) { sub_y += den; --cy; og_apply(g, cx, cy); } }
        ++cx;
        if (up) { if (sub_y == den) { sub_y -= den; ++cy; } }
        else { if (sub_y == 0) { sub_y += den; --cy; } }
        if (cx == end_x) break;
        sub_y += dy * 2 * S;
    }
    sub_y += dy * last_pixel;
    og_apply(g, cx, cy);
    if (up) { while (sub_y > den) { sub_y -= den; ++cy; og_apply(g, cx, cy); } }
    else { while (sub_y < 0) { sub_y += den; --cy; og_apply(g, cx, cy); } }
}

int ogrid_insert(uint16_t *cells, int nx, int ny, double resolution, double max_x, double max_y, const float origin[2],
                 const float *returns_xy, int n_ret, const float *misses_xy, int n_miss, float hit_probability,
                 float miss_probability, int insert_free_space)
{
    uint16_t *hit = (uint16_t *)malloc(2 * 32768), *miss = (uint16_t *)malloc(2 * 32768);
    int *ends = (int *)malloc(sizeof(int) * 2 * (size_t)(n_ret + n_miss + 1));
    ogrid_lookup_table(hit_probability, hit);
    ogrid_lookup_table(miss_probability, miss);
    const double rs = resolution / OG_SUBPIXEL;                                  /* superscaled limits (:48-53) */
    const long long sx = (long long)nx * OG_SUBPIXEL, sy = (long long)ny * OG_SUBPIXEL;
    int rc = 0;
#define OG_IDX(px, py, ox, oy) do { ox = (int)lround((max_y - (double)(py)) / rs - 0.5); oy = (int)lround((max_x - (double)(px)) / rs - 0.5); \
                                    if (ox < 0 || oy < 0 || ox >= sx || oy >= sy) rc = -1; } while (0)
    int bx, by;
    OG_IDX(origin[0], origin[1], bx, by);
    for (int i = 0; i < n_ret; ++i) OG_IDX(returns_xy[2 * i], returns_xy[2 * i + 1], ends[2 * i], ends[2 * i + 1]);
    for (int i = 0; i < n_miss; ++i) OG_IDX(misses_xy[2 * i], misses_xy[2 * i + 1], ends[2 * (n_ret + i)], ends[2 * (n_ret + i) + 1]);
    if (rc == 0) {
        og_apply_ctx g = {cells, nx, ny, hit};
        for (int i = 0; i < n_ret; ++i) og_apply(&g, ends[2 * i] / OG_SUBPIXEL, ends[2 * i + 1] / OG_SUBPIXEL);   /* hits first (:57-62) */
        if (insert_free_space) {
            g.table = miss;
            for (int i = 0; i < n_ret + n_miss; ++i) og_ray(&g, bx, by, ends[2 * i], ends[2 * i + 1]);          /* (:69-91) */
        }
        for (size_t k = 0; k < (size_t)nx * ny; ++k) if (cells[k] >= OG_MARKER) cells[k] -= OG_MARKER;         /* FinishUpdate */
    }
    free(hit); free(miss); free(ends);
    return rc;
}

/* ------------------------------------------------------------------------------------------------------------
 * GrowAsNeeded + Grid2D::GrowLimits (probability_grid_range_data_inserter_2d.cc:20-38, grid_2d.cc:59-99).
 * Limits only: `dims` = (num_x_cells, num_y_cells) and `max_xy` are updated in place, `offset` receives where
 * the old cell (0, 0) lands in the grown grid.  The cell copy is ogrid_grow_copy. */
static void og_grow_point(int dims[2], double resolution, double max_xy[2], float px, float py, int offset[2])
{
    for (;;) {
        const long ix = lround((max_xy[1] - (double)py) / resolution - 0.5);       /* MapLimits::GetCellIndex (map_limits.h:48-57) */
        const long iy = lround((max_xy[0] - (double)px) / resolution - 0.5);
        if (ix >= 0 && iy >= 0 && ix < dims[0] && iy < dims[1]) return;            /* Contains (map_limits.h:67-73) */
        const int x_offset = dims[0] / 2, y_offset = dims[1] / 2;                  /* grid_2d.cc:66-67 */
        max_xy[0] = max_xy[0] + resolution * (double)y_offset;                     /* (:70-71): max + res * (y_offset, x_offset) */
        max_xy[1] = max_xy[1] + resolution * (double)x_offset;
        dims[0] *= 2; dims[1] *= 2;
        offset[0] += x_offset; offset[1] += y_offset;
    }
}
void ogrid_grow_limits(int dims[2], double resolution, double max_xy[2], const float origin[2], const float *returns_xy,
                       int n_ret, const float *misses_xy, int n_miss, int offset[2])
{
    float lo[2] = {origin[0], origin[1]}, hi[2] = {origin[0], origin[1]};          /* AlignedBox2f(origin), extend (:23-33) */
    for (int i = 0; i < n_ret; ++i)
        for (int k = 0; k < 2; ++k) { const float v = returns_xy[2 * i + k]; if (v < lo[k]) lo[k] = v; if (v > hi[k]) hi[k] = v; }
    for (int i = 0; i < n_miss; ++i)
        for (int k = 0; k < 2; ++k) { const float v = misses_xy[2 * i + k]; if (v < lo[k]) lo[k] = v; if (v > hi[k]) hi[k] = v; }
    const float pad = 1e-6f;
    offset[0] = offset[1] = 0;
    og_grow_point(dims, resolution, max_xy, lo[0] - pad, lo[1] - pad, offset);     /* (:34-37) */
    og_grow_point(dims, resolution, max_xy, hi[0] + pad, hi[1] + pad, offset);
}
/* new_cells (nnx * nny) = unknown everywhere, the old grid at `offset` (grid_2d.cc:81-91) */
void ogrid_grow_copy(const uint16_t *cells, int nx, int ny, uint16_t *new_cells, int nnx, int nny, const int offset[2])
{
    memset(new_cells, 0, sizeof(uint16_t) * (size_t)nnx * nny);
    for (int i = 0; i < ny; ++i)
        for (int j = 0; j < nx; ++j) new_cells[(size_t)(offset[0] + j) + (size_t)(offset[1] + i) * nnx] = cells[j + (size_t)i * nx];
}

/* ------------------------------------------------------------------------------------------------------------
 * CeresScanMatcher2D::Match (src/scan_matching/ceres_scan_matcher_2d.cc:26-62): three residual blocks on the pose
 * (x, y, angle) --
 *   occupied space (occupied_space_cost_function_2d.cc:25-52): per point, weight / sqrt(n) * the bicubic
 *     interpolation of the correspondence cost at the point's (row, column) = ((max.x - wx) / res - 0.5,
 *     (max.y - wy) / res - 0.5), both shifted by kPadding = INT_MAX / 4; outside the grid the cost is
 *     kMaxCorrespondenceCost (:69-81)
 *   translation delta (translation_delta_cost_functor_2d.h:24-29), rotation delta (rotation_delta_cost_functor_2d.h:24-28)
 * minimised by ceres::Solve with the options of src/ros_node.cc:364-377 (DENSE_QR, use_nonmonotonic_steps,
 * max_num_iterations 100; everything else Ceres' defaults).
 *
 * Ceres Solver is a third-party dependency that is absent here and un-pinned by the reference
 * (CMakeLists.txt:34 `find_package(Ceres REQUIRED ...)`): PARITY UNPINNED.  What follows restates its published
 * algorithm -- ceres::BiCubicInterpolator / CubicHermiteSpline (cubic_interpolation.h: Catmull-Rom, first along the
 * columns of the four rows, then along the rows), automatic differentiation replaced by the analytic chain rule, and
 * the Levenberg-Marquardt trust-region loop of trust_region_minimizer.cc / levenberg_marquardt_strategy.cc /
 * trust_region_step_evaluator.cc (1.13 and later): Jacobi scaling 1 / (1 + ||column||) fixed at iteration 0, LM
 * diagonal clamp(diag(J'J), 1e-6, 1e32) / radius, model cost change -(J d)'(r + J d / 2), parameter tolerance 1e-8,
 * function tolerance 1e-6, gradient tolerance 1e-10, min relative decrease 1e-3, radius 1e4 .. 1e16 (min 1e-32),
 * radius update r / max(1/3, 1 - (2 rho - 1)^3) on success and r / 2, / 4, ... on failure, Conn-Gould-Toint
 * non-monotonic acceptance over 5 steps, the answer = the accepted iterate of least cost.  The 3-parameter damped
 * least-squares step is solved through its normal equations (Cholesky) instead of a QR of the stacked Jacobian:
 * the same minimiser up to rounding. */
#define OG_PAD 536870911.0                                     /* kPadding = INT_MAX / 4 (:57) */
typedef struct {
    const uint16_t *cells; int nx, ny; double res, max_x, max_y;
    const float *pts; int n;
    double w_occ, w_t, w_r, tx, ty, angle0;
} og_refine_ctx;

static double og_cost_at(const og_refine_ctx *c, long row, long col)             /* GridArrayAdapter::GetValue (:69-81) */
{
    const long y = row - (long)OG_PAD, x = col - (long)OG_PAD;
    if (x < 0 || y < 0 || x >= c->nx || y >= c->ny) return (double)0.9f;        /* kMaxCorrespondenceCost */
    return (double)og_value_to_cost(c->cells[(size_t)c->nx * y + x]);
}
static void og_hermite(double p0, double p1, double p2, double p3, double x, double *f, double *dfdx)
{
    const double a = 0.5 * (-p0 + 3.0 * p1 - 3.0 * p2 + p3);
    const double b = 0.5 * (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3);
    const double cc = 0.5 * (-p0 + p2);
    const double d = p1;
    if (f) *f = d + x * (cc + x * (b + x * a));
    if (dfdx) *dfdx = cc + x * (2.0 * b + 3.0 * a * x);
}
/* cost = 1/2 |r|^2, g = J'r, H = J'J (upper triangle xx, xy, xt, yy, yt, tt) at pose p */
static void og_refine_eval(const og_refine_ctx *c, const double p[3], double *cost, double g[3], double H[6])
{
    const double cs = cos(p[2]), sn = sin(p[2]);
    const double scale = c->w_occ / sqrt((double)c->n);
    double ss = 0.;
    for (int k = 0; k < 3; ++k) g[k] = 0.;
    for (int k = 0; k < 6; ++k) H[k] = 0.;
    for (int i = 0; i < c->n; ++i) {
        const double px = (double)c->pts[2 * i], py = (double)c->pts[2 * i + 1];
        const double wx = cs * px - sn * py + p[0], wy = sn * px + cs * py + p[1];
        const double dwx = -sn * px - cs * py, dwy = cs * px - sn * py;          /* d world / d angle */
        const double r = (c->max_x - wx) / c->res - 0.5 + OG_PAD, q = (c->max_y - wy) / c->res - 0.5 + OG_PAD;
        const double rf = floor(r), qf = floor(q);
        const long row = (long)rf, col = (long)qf;
        double f[4], dq[4];
        for (int a = 0; a < 4; ++a)
            og_hermite(og_cost_at(c, row - 1 + a, col - 1), og_cost_at(c, row - 1 + a, col), og_cost_at(c, row - 1 + a, col + 1),
                       og_cost_at(c, row - 1 + a, col + 2), q - qf, &f[a], &dq[a]);
        double v, dvdr, dvdq;
        og_hermite(f[0], f[1], f[2], f[3], r - rf, &v, &dvdr);
        og_hermite(dq[0], dq[1], dq[2], dq[3], r - rf, &dvdq, NULL);
        const double res_i = scale * v;
        const double ninv = -1.0 / c->res;                                       /* Jet / scalar multiplies by 1 / scalar (ceres/jet.h) */
        const double J[3] = {scale * (dvdr * ninv), scale * (dvdq * ninv), scale * (dvdr * (dwx * ninv) + dvdq * (dwy * ninv))};
        ss += res_i * res_i;
        for (int k = 0; k < 3; ++k) g[k] += J[k] * res_i;
        H[0] += J[0] * J[0]; H[1] += J[0] * J[1]; H[2] += J[0] * J[2]; H[3] += J[1] * J[1]; H[4] += J[1] * J[2]; H[5] += J[2] * J[2];
    }
    const double r0 = c->w_t * (p[0] - c->tx), r1 = c->w_t * (p[1] - c->ty), r2 = c->w_r * (p[2] - c->angle0);
    ss += r0 * r0 + r1 * r1 + r2 * r2;
    g[0] += c->w_t * r0; g[1] += c->w_t * r1; g[2] += c->w_r * r2;
    H[0] += c->w_t * c->w_t; H[3] += c->w_t * c->w_t; H[5] += c->w_r * c->w_r;
    *cost = 0.5 * ss;
}
/* (A) y = b for the symmetric positive definite 3x3 A (upper triangle); 0 on failure */
static int og_chol3(const double A[6], const double b[3], double y[3])
{
    const double l00 = sqrt(A[0]);
    if (!(l00 > 0.)) return 0;
    const double l10 = A[1] / l00, l20 = A[2] / l00;
    const double d1 = A[3] - l10 * l10;
    if (!(d1 > 0.)) return 0;
    const double l11 = sqrt(d1), l21 = (A[4] - l20 * l10) / l11;
    const double d2 = A[5] - l20 * l20 - l21 * l21;
    if (!(d2 > 0.)) return 0;
    const double l22 = sqrt(d2);
    const double z0 = b[0] / l00, z1 = (b[1] - l10 * z0) / l11, z2 = (b[2] - l20 * z0 - l21 * z1) / l22;
    y[2] = z2 / l22; y[1] = (z1 - l21 * y[2]) / l11; y[0] = (z0 - l10 * y[1] - l20 * y[2]) / l00;
    return isfinite(y[0]) && isfinite(y[1]) && isfinite(y[2]);
}
/* options = {occupied_space_weight, translation_weight, rotation_weight, max_num_iterations, use_nonmonotonic_steps};
 * summary = {initial_cost, final_cost, iterations, termination (0 convergence, 1 no convergence, 2 failure)} */
int ogrid_refine_match(const double options[5], const double target_translation[2], const double initial_pose[3],
                       const float *points_xy, int n, const uint16_t *cells, int nx, int ny, double resolution, double max_x,
                       double max_y, double pose_estimate[3], double summary[4])
{
    if (n <= 0) return -1;
    og_refine_ctx c = {cells, nx, ny, resolution, max_x, max_y, points_xy, n, options[0], options[1], options[2],
                       target_translation[0], target_translation[1], initial_pose[2]};
    const int max_iter = (int)options[3], max_nonmono = options[4] != 0. ? 5 : 0;
    double x[3] = {initial_pose[0], initial_pose[1], initial_pose[2]}, x_cost, g[3], H[6];
    og_refine_eval(&c, x, &x_cost, g, H);                                       /* IterationZero */
    double s[3] = {1. / (1. + sqrt(H[0])), 1. / (1. + sqrt(H[3])), 1. / (1. + sqrt(H[5]))};
    double x_norm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    double gmax = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
    double radius = 1e4, decrease = 2.;
    double ev_min = x_cost, ev_cur = x_cost, ev_ref = x_cost, ev_cand = x_cost, acc_ref = 0., acc_cand = 0.;
    int nonmono = 0, invalid = 0, iter = 0, successful = 1, termination = 1;
    double best[3] = {x[0], x[1], x[2]}, min_cost = INFINITY;
    summary[0] = x_cost;
    for (;;) {
        if (successful && x_cost < min_cost) { min_cost = x_cost; best[0] = x[0]; best[1] = x[1]; best[2] = x[2]; }
        if (iter >= max_iter) { termination = 1; break; }
        if (successful && gmax <= 1e-10) { termination = 0; break; }
        if (radius < 1e-32) { termination = 0; break; }
        ++iter;
        /* LevenbergMarquardtStrategy::ComputeStep on the column-scaled Jacobian */
        const double Hs[6] = {s[0] * H[0] * s[0], s[0] * H[1] * s[1], s[0] * H[2] * s[2], s[1] * H[3] * s[1], s[1] * H[4] * s[2], s[2] * H[5] * s[2]};
        const double gs[3] = {s[0] * g[0], s[1] * g[1], s[2] * g[2]};
        double A[6] = {Hs[0], Hs[1], Hs[2], Hs[3], Hs[4], Hs[5]}, y[3], step[3];
        A[0] += fmin(fmax(Hs[0], 1e-6), 1e32) / radius; A[3] += fmin(fmax(Hs[3], 1e-6), 1e32) / radius; A[5] += fmin(fmax(Hs[5], 1e-6), 1e32) / radius;
        double mcc = -1.;
        if (og_chol3(A, gs, y)) {
            for (int k = 0; k < 3; ++k) step[k] = -y[k];
            const double Hd[3] = {Hs[0] * step[0] + Hs[1] * step[1] + Hs[2] * step[2], Hs[1] * step[0] + Hs[3] * step[1] + Hs[4] * step[2],
                                  Hs[2] * step[0] + Hs[4] * step[1] + Hs[5] * step[2]};
            mcc = -(step[0] * gs[0] + step[1] * gs[1] + step[2] * gs[2]) - 0.5 * (step[0] * Hd[0] + step[1] * Hd[1] + step[2] * Hd[2]);
        }
        if (!(mcc > 0.)) {                                                       /* HandleInvalidStep */
            successful = 0;
            if (++invalid >= 5) { termination = 2; break; }
            radius /= decrease; decrease *= 2.;
            continue;
        }
        invalid = 0;
        const double xc[3] = {x[0] + step[0] * s[0], x[1] + step[1] * s[1], x[2] + step[2] * s[2]};
        double c_cost, cg[3], cH[6];
        og_refine_eval(&c, xc, &c_cost, cg, cH);
        const double d0 = x[0] - xc[0], d1 = x[1] - xc[1], d2 = x[2] - xc[2];
        if (sqrt(d0 * d0 + d1 * d1 + d2 * d2) <= 1e-8 * (x_norm + 1e-8)) { termination = 0; break; }      /* ParameterToleranceReached */
        if (fabs(x_cost - c_cost) <= 1e-6 * x_cost) { termination = 0; break; }                            /* FunctionToleranceReached */
        const double rho = fmax((ev_cur - c_cost) / mcc, (ev_ref - c_cost) / (acc_ref + mcc));             /* StepQuality */
        if (rho > 1e-3) {                                                        /* HandleSuccessfulStep */
            for (int k = 0; k < 3; ++k) { x[k] = xc[k]; g[k] = cg[k]; }
            for (int k = 0; k < 6; ++k) H[k] = cH[k];
            x_norm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
            x_cost = c_cost;
            gmax = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
            successful = 1;
            const double t = 2. * rho - 1.;
            radius = fmin(1e16, radius / fmax(1. / 3., 1. - t * t * t));
            decrease = 2.;
            ev_cur = c_cost; acc_cand += mcc; acc_ref += mcc;                    /* TrustRegionStepEvaluator::StepAccepted */
            if (ev_cur < ev_min) { ev_min = ev_cur; nonmono = 0; ev_cand = ev_cur; acc_cand = 0.; }
            else { ++nonmono; if (ev_cur > ev_cand) { ev_cand = ev_cur; acc_cand = 0.; } }
            if (nonmono == max_nonmono) { ev_ref = ev_cand; acc_ref = acc_cand; }
        } else {                                                                 /* HandleUnsuccessfulStep */
            successful = 0;
            radius /= decrease; decrease *= 2.;
        }
    }
    pose_estimate[0] = best[0]; pose_estimate[1] = best[1]; pose_estimate[2] = best[2];
    summary[1] = min_cost; summary[2] = (double)iter; summary[3] = (double)termination;
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------
 * ProbabilityGrid::DrawToSubmapTexture (src/mapping/probability_grid.cc:86-131) without the gzip container:
 * the grid cropped to the bounding box of its known cells (Grid2D::ComputeCroppedLimits, grid_2d.cc:36-48 -- the
 * box ApplyLookupTable keeps is the box of the cells that are not 0), two bytes per cell in x-fastest order
 * (xy_index.h:45-60): (value, alpha) from 128 - ProbabilityToLogOddsInteger(probability) (submaps.h:22-41).
 * box = (offset_x, offset_y, width, height); slice_max = limits.max - resolution * (offset_y, offset_x) (:122-123).
 * Returns the number of bytes written (2 * width * height), or -1 if `cap` is too small. */
static void og_texture_bytes(uint16_t v, uint8_t out[2])
{
    if (v == 0) { out[0] = 0; out[1] = 0; return; }                             /* unknown (:97-102) */
    const float kMinP = 0.1f, kMaxP = 1.f - kMinP;
    const float kMaxLogOdds = logf(kMaxP / (1.f - kMaxP)), kMinLogOdds = logf(kMinP / (1.f - kMinP));
    const float p = 1.f - og_value_to_cost(v);                                  /* GetProbability (:57-63) */
    const float logit = logf(p / (1.f - p));
    const int li = (int)lroundf((logit - kMinLogOdds) * 254.f / (kMaxLogOdds - kMinLogOdds)) + 1;
    const int delta = 128 - li;
    const uint8_t alpha = (uint8_t)(delta > 0 ? 0 : -delta), value = (uint8_t)(delta > 0 ? delta : 0);
    out[0] = value;
    out[1] = (value || alpha) ? alpha : 1;
}
long ogrid_draw_texture(const uint16_t *cells, int nx, int ny, double resolution, double max_x, double max_y, uint8_t *out,
                        long cap, int box[4], double slice_max[2])
{
    int x0 = nx, y0 = ny, x1 = -1, y1 = -1;
    for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x)
            if (cells[(size_t)nx * y + x] != 0) { if (x < x0) x0 = x; if (x > x1) x1 = x; if (y < y0) y0 = y; if (y > y1) y1 = y; }
    if (x1 < 0) { x0 = y0 = 0; x1 = y1 = 0; }                                   /* empty box: offset 0, CellLimits(1, 1) (grid_2d.cc:39-44) */
    const int w = x1 - x0 + 1, hgt = y1 - y0 + 1;
    box[0] = x0; box[1] = y0; box[2] = w; box[3] = hgt;
    slice_max[0] = max_x - resolution * y0;
    slice_max[1] = max_y - resolution * x0;
    if (2l * w * hgt > cap) return -1;
    for (int y = 0; y < hgt; ++y)
        for (int x = 0; x < w; ++x) og_texture_bytes(cells[(size_t)nx * (y0 + y) + (x0 + x)], out + 2 * ((size_t)w * y + x));
    return 2l * w * hgt;
}
