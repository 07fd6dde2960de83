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
float ogrid_value_to_probability(uint16_t value)
{
    const float kMinProbability = 0.1f, kMaxProbability = 1.f - kMinProbability;
    const float lower = 1.f - kMaxProbability /* kMinCorrespondenceCost */, upper = 1.f - kMinProbability /* kMax... */;
    const uint16_t v = (uint16_t)(value & 32767u);             /* the table repeats for values with the update marker */
    float cost;
    if (v == 0) cost = upper;                                   /* unknown_result = kMaxCorrespondenceCost */
    else {
        const float kScale = (upper - lower) / (32768 - 2.f);
        cost = v * kScale + (lower - kScale);
    }
    return 1.f - cost;                                          /* CorrespondenceCostToProbability */
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
 * GrowAsNeeded / GrowLimits (:19-38, grid_2d.cc:47-98) is NOT restated: the caller hands in a grid that already
 * contains the origin and every end point; anything outside is an error (return -1) and leaves the grid untouched. */
#define OG_MARKER 32768u
#define OG_SUBPIXEL 1000

static float og_value_to_cost(int v)          /* kValueToCorrespondenceCost, v in [0, 32767] (probability_values.cc:11-20,46-51) */
{
    const float kMinProbability = 0.1f, kMaxProbability = 1.f - kMinProbability;
    const float lower = 1.f - kMaxProbability, upper = 1.f - kMinProbability;
    if (v == 0) return upper;
    const float kScale = (upper - lower) / (32768 - 2.f);
    return v * kScale + (lower - kScale);
}
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
