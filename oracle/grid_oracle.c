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
