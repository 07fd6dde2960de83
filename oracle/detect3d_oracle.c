/*
 * oracle/detect3d_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's 3D reflector detector,
 * src/reflector_detect/point_cloud/point_cloud_reflector_detect.cc:9-106.  Nearly all of its
 * arithmetic lives in PCL 1.7 (CMakeLists.txt:41), which is neither vendored under
 * /root/reference nor installed here, so the PCL stages are restated from PCL's published
 * algorithms (file names below are PCL 1.7's):
 *
 *   StatisticalOutlierRemoval (filters/impl/statistical_outlier_removal.hpp, applyFilterIndices)
 *     per point: nearestKSearch(k = MeanK + 1) -- the query itself comes back first and is
 *     skipped; distance_i = float( sum_{k=1..MeanK} sqrt(d2_k) / MeanK ) with d2 float32
 *     (FLANN L2_Simple), the sum in double; a point whose search returns fewer than MeanK+1
 *     neighbours gets distance 0 and is not counted.  mean and (n-1)-variance in double over
 *     the counted points; keep distance <= mean + StddevMulThresh * stddev.
 *   EuclideanClusterExtraction (segmentation/impl/extract_clusters.hpp, extractEuclideanClusters)
 *     seeded region growing in point order with radiusSearch(tolerance) (FLANN: squared
 *     distance < tolerance^2); a grown set of size in [min, max] becomes a cluster, its indices
 *     sorted ascending; clusters are finally sorted by size, largest first.
 *   compute3DCentroid (common/impl/centroid.hpp): float32 running sum in index order / count.
 *
 * PARITY UNPINNED (stated in DESIGN.md): no PCL, no reference tests, and PCL's final sort is
 * unstable, so the order of equal-size clusters is unspecified in the reference.  This oracle
 * fixes it: size descending, then smallest member index ascending.  Exact k-NN ties (equal
 * float32 distances at rank MeanK+1) are likewise resolved by value only (the sum of the
 * MeanK+1 smallest distances does not depend on which tied point is taken).
 *
 * Second witness: tests/witness/detect3d_witness.py (scipy kd-tree + connected components) must give the
 * same centres bit for bit (tests/test_witness_cpu.py).
 *
 * Compile with -ffp-contract=off.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define O3_MEAN_K 30              /* point_cloud_reflector_detect.cc:45 */
#define O3_STD_MUL 0.5            /* :46 */
#define O3_TOL 0.2                /* :69 */
#define O3_MIN 4                  /* :70 */
#define O3_MAX 160                /* :71 */

static float d2f(const float *a, const float *b)
{
    const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    float r = dx * dx;            /* FLANN L2_Simple: float accumulation over the dimensions */
    r += dy * dy;
    r += dz * dz;
    return r;
}

/*
 * HandlePointCloud.  xyzi: N x 4 float32 (x, y, z, intensity).  s2b = Project2D(sensor_to_base_link)
 * (x, y, yaw).  Writes up to max_centers centres; returns K, or -2 if max_centers is too small.
 * Optional outputs (may be NULL): n_after_intensity, n_after_sor.
 */
int od3_handle_cloud_ex(double intensity_min, const double s2b[3], const float *xyzi, int N,
                        float *centers_xy, int max_centers, int *n_after_intensity, int *n_after_sor)
{
    /* ---- intensity filter (:31-39) */
    float *p = (float *)malloc(sizeof(float) * 3 * (size_t)(N > 0 ? N : 1));
    int M = 0;
    for (int i = 0; i < N; ++i)
        if ((double)xyzi[4 * i + 3] > intensity_min) {           /* :33 float > double */
            p[3 * M] = xyzi[4 * i]; p[3 * M + 1] = xyzi[4 * i + 1]; p[3 * M + 2] = xyzi[4 * i + 2];
            ++M;
        }
    if (n_after_intensity) *n_after_intensity = M;

    /* ---- StatisticalOutlierRemoval (:43-47) */
    float *dist = (float *)calloc((size_t)(M > 0 ? M : 1), sizeof(float));
    int valid = 0;
    if (M >= O3_MEAN_K + 1) {
        float best[O3_MEAN_K + 1];
        for (int i = 0; i < M; ++i) {
            int nb = 0;
            for (int j = 0; j < M; ++j) {                            /* exact k-NN by brute force */
                const float d2 = d2f(p + 3 * i, p + 3 * j);
                if (nb < O3_MEAN_K + 1) {
                    int q = nb++;
                    while (q > 0 && best[q - 1] > d2) { best[q] = best[q - 1]; --q; }
                    best[q] = d2;
                } else if (d2 < best[O3_MEAN_K]) {
                    int q = O3_MEAN_K;
                    while (q > 0 && best[q - 1] > d2) { best[q] = best[q - 1]; --q; }
                    best[q] = d2;
                }
            }
            double dist_sum = 0;
            for (int k = 1; k < O3_MEAN_K + 1; ++k) dist_sum += sqrtf(best[k]);   /* k = 0 is the query */
            dist[i] = (float)(dist_sum / O3_MEAN_K);
            ++valid;
        }
    }
    double sum = 0, sq_sum = 0;
    for (int i = 0; i < M; ++i) { sum += dist[i]; sq_sum += (double)dist[i] * dist[i]; }
    const double mean = sum / (double)valid;
    const double variance = (sq_sum - sum * sum / (double)valid) / ((double)valid - 1);
    const double thr = mean + O3_STD_MUL * sqrt(variance);
    float *q = (float *)malloc(sizeof(float) * 3 * (size_t)(M > 0 ? M : 1));
    int M2 = 0;
    for (int i = 0; i < M; ++i)
        if (!(dist[i] > thr)) {                                       /* NaN threshold keeps everything */
            memcpy(q + 3 * M2, p + 3 * i, sizeof(float) * 3);
            ++M2;
        }
    if (n_after_sor) *n_after_sor = M2;

    /* ---- EuclideanClusterExtraction (:65-74) */
    char *processed = (char *)calloc((size_t)(M2 > 0 ? M2 : 1), 1);
    int *queue = (int *)malloc(sizeof(int) * (size_t)(M2 > 0 ? M2 : 1));
    int *label = (int *)malloc(sizeof(int) * (size_t)(M2 > 0 ? M2 : 1));   /* cluster id per point, -1 none */
    int *csize = (int *)malloc(sizeof(int) * (size_t)(M2 > 0 ? M2 : 1));
    int *cmin = (int *)malloc(sizeof(int) * (size_t)(M2 > 0 ? M2 : 1));
    int nc = 0;
    const float tol2 = (float)(O3_TOL * O3_TOL);
    for (int i = 0; i < M2; ++i) label[i] = -1;
    for (int i = 0; i < M2; ++i) {
        if (processed[i]) continue;
        int qn = 0, qi = 0;
        queue[qn++] = i; processed[i] = 1;
        while (qi < qn) {
            const int s = queue[qi++];
            for (int j = 0; j < M2; ++j)
                if (!processed[j] && d2f(q + 3 * s, q + 3 * j) < tol2) { queue[qn++] = j; processed[j] = 1; }
        }
        if (qn >= O3_MIN && qn <= O3_MAX) {
            int mn = queue[0];
            for (int k = 0; k < qn; ++k) { label[queue[k]] = nc; if (queue[k] < mn) mn = queue[k]; }
            csize[nc] = qn; cmin[nc] = mn; ++nc;
        }
    }
    /* order: size descending, then smallest member index ascending (our fixed total order) */
    int *order = (int *)malloc(sizeof(int) * (size_t)(nc > 0 ? nc : 1));
    for (int c = 0; c < nc; ++c) order[c] = c;
    for (int a = 1; a < nc; ++a) {
        const int c = order[a];
        int b = a;
        while (b > 0 && (csize[order[b - 1]] < csize[c] || (csize[order[b - 1]] == csize[c] && cmin[order[b - 1]] > cmin[c]))) {
            order[b] = order[b - 1]; --b;
        }
        order[b] = c;
    }
    int K = nc;
    if (K > max_centers) K = -2;
    else {
        const float sx = (float)s2b[0], sy = (float)s2b[1], sa = (float)s2b[2];
        const float cs = cosf(sa), sn = sinf(sa);
        for (int r = 0; r < nc; ++r) {
            const int c = order[r];
            float cx = 0.f, cy = 0.f, cz = 0.f;
            for (int i = 0; i < M2; ++i)                                 /* ascending index order */
                if (label[i] == c) { cx += q[3 * i]; cy += q[3 * i + 1]; cz += q[3 * i + 2]; }
            cx /= (float)csize[c]; cy /= (float)csize[c]; cz /= (float)csize[c];   /* :94 */
            (void)cz;                                                    /* z is dropped (:95, Q16) */
            centers_xy[2 * r] = (cs * cx + (-sn) * cy) + sx;             /* :96 Rigid2f * point */
            centers_xy[2 * r + 1] = (sn * cx + cs * cy) + sy;
        }
    }
    free(p); free(dist); free(q); free(processed); free(queue); free(label); free(csize); free(cmin); free(order);
    return K;
}

int od3_handle_cloud(double intensity_min, const double s2b[3], const float *xyzi, int N,
                     float *centers_xy, int max_centers)
{
    return od3_handle_cloud_ex(intensity_min, s2b, xyzi, N, centers_xy, max_centers, 0, 0);
}
