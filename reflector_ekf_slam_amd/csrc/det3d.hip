// det3d.hip -- MI355X-native 3D (point cloud) reflector detector behind include/rdet.h.
//
// Replaces reflector_detect::PointCloudReflectorDetect::HandlePointCloud (reference
// src/reflector_detect/point_cloud/point_cloud_reflector_detect.cc:9-106), whose arithmetic
// is PCL 1.7's (un-vendored; the PCL semantics this file implements are spelled out in DESIGN.md section 4):
//
//   k3_filter     intensity > threshold, order-preserving compaction            (:31-39)
//   k3_knn        StatisticalOutlierRemoval part 1: per point the MeanK+1 = 31 smallest float32
//                 squared distances (the query first), brute force over LDS-staged candidate
//                 tiles, one lane per point with its sorted list in a private LDS column   (:43-47)
//   k3_sor        part 2: mean / (n-1)-variance in FP64, threshold, second compaction
//   k3_propagate  EuclideanClusterExtraction as connected components of the radius-0.2 m graph:
//                 in-place min-label propagation with pointer jumping, launched a fixed number
//                 of rounds that exit immediately once a round changed nothing        (:65-74)
//   k3_finish     component sizes, size gate [4,160], order (size desc, first index asc),
//                 float32 centroids in index order, Rigid2f to base_link              (:77-97)
//
// No kd-tree: after the intensity gate a cloud holds 10^2..10^4 points, for which the
// all-pairs distance sweep is a coalesced, LDS-tiled, embarrassingly parallel kernel, while a
// tree build would be pointer-chasing.  Nothing waits on the host between stages: the point
// counts M, M2 stay on the device and every grid is sized for the capacity.
#include "../../include/rdet.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

namespace {

constexpr int MEAN_K = 30;          // point_cloud_reflector_detect.cc:45
constexpr double STD_MUL = 0.5;     // :46
constexpr float TOL2 = (float)(0.2 * 0.2);   // :69 (FLANN radius search: squared distance < r^2)
constexpr int MIN_SZ = 4, MAX_SZ = 160;      // :70-71
constexpr int ROUNDS = 64;

struct Det3dCtl {
    int M, M2, K, err;
    int changed[ROUNDS + 1];
    float centers[2 * RDET_MAX_CENTERS];
};

struct Det3dBufs {
    const float *xyzi;
    float *p1;        // 3 x cap, SoA: x | y | z  after the intensity filter
    float *p2;        // after SOR
    float *dist;      // SOR mean neighbour distance
    int *label;
    int *cnt;
    Det3dCtl *ctl;
    int cap;
};

__device__ static float d2f(float ax, float ay, float az, float bx, float by, float bz)
{
#pragma clang fp contract(off)
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    float r = dx * dx;           // FLANN L2_Simple: float accumulation over x, y, z
    r += dy * dy;
    r += dz * dz;
    return r;
}

__device__ static int block_excl_sum(int v, int *lds, int *total)
{
    const int tid = threadIdx.x;
    lds[tid] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int t = (tid >= off) ? lds[tid - off] : 0;
        __syncthreads();
        lds[tid] += t;
        __syncthreads();
    }
    const int incl = lds[tid];
    if (total) *total = lds[1023];
    __syncthreads();
    return incl - v;
}

// ---- intensity filter + compaction (one workgroup keeps the point order) ---------------------
__global__ __launch_bounds__(1024) void k3_filter(Det3dBufs B, int N, double intensity_min)
{
    __shared__ int lds[1024];
    const int tid = threadIdx.x;
    const int CH = (N + 1023) / 1024;
    const int b0 = tid * CH, b1 = min(N, b0 + CH);
    int c = 0;
    for (int i = b0; i < b1; ++i) c += ((double)B.xyzi[4 * i + 3] > intensity_min) ? 1 : 0;   // :33
    int M;
    int pos = block_excl_sum(c, lds, &M);
    for (int i = b0; i < b1; ++i)
        if ((double)B.xyzi[4 * i + 3] > intensity_min) {
            B.p1[pos] = B.xyzi[4 * i];
            B.p1[B.cap + pos] = B.xyzi[4 * i + 1];
            B.p1[2 * B.cap + pos] = B.xyzi[4 * i + 2];
            ++pos;
        }
    if (tid == 0) {
        B.ctl->M = M; B.ctl->M2 = 0; B.ctl->K = 0; B.ctl->err = 0;
        for (int r = 0; r <= ROUNDS; ++r) B.ctl->changed[r] = 0;
    }
}

// ---- SOR part 1: mean distance to the MeanK nearest neighbours ---------------------------------
__global__ __launch_bounds__(256) void k3_knn(Det3dBufs B)
{
    __shared__ float best[MEAN_K + 1][256];       // lane-private sorted lists, one LDS column each
    __shared__ float tx[256], ty[256], tz[256];
    const int M = B.ctl->M;
    if ((int)(blockIdx.x * 256) >= M) return;
    const int tid = threadIdx.x;
    const int i = blockIdx.x * 256 + tid;
    const bool live = i < M;
    const float *X = B.p1, *Y = B.p1 + B.cap, *Z = B.p1 + 2 * B.cap;
    const float px = live ? X[i] : 0.f, py = live ? Y[i] : 0.f, pz = live ? Z[i] : 0.f;
    int nb = 0;
    for (int j0 = 0; j0 < M; j0 += 256) {
        const int j = j0 + tid;
        __syncthreads();
        if (j < M) { tx[tid] = X[j]; ty[tid] = Y[j]; tz[tid] = Z[j]; }
        __syncthreads();
        const int jn = min(256, M - j0);
        if (live)
            for (int t = 0; t < jn; ++t) {
                const float d2 = d2f(px, py, pz, tx[t], ty[t], tz[t]);
                if (nb < MEAN_K + 1) {
                    int q = nb++;
                    while (q > 0 && best[q - 1][tid] > d2) { best[q][tid] = best[q - 1][tid]; --q; }
                    best[q][tid] = d2;
                } else if (d2 < best[MEAN_K][tid]) {
                    int q = MEAN_K;
                    while (q > 0 && best[q - 1][tid] > d2) { best[q][tid] = best[q - 1][tid]; --q; }
                    best[q][tid] = d2;
                }
            }
    }
    if (live) {
        float dst = 0.f;                                      // search "failed": fewer than MeanK+1 points
        if (M >= MEAN_K + 1) {
            double dist_sum = 0;
            for (int k = 1; k < MEAN_K + 1; ++k) dist_sum += sqrtf(best[k][tid]);   // k = 0 is the query itself
            dst = (float)(dist_sum / MEAN_K);
        }
        B.dist[i] = dst;
    }
}

// ---- SOR part 2: statistics, threshold, second compaction ---------------------------------------
__global__ __launch_bounds__(1024) void k3_sor(Det3dBufs B)
{
    __shared__ int lds[1024];
    __shared__ double red[2][1024];
    __shared__ double s_thr;
    const int tid = threadIdx.x;
    const int M = B.ctl->M;
    const int CH = (M + 1023) / 1024;
    const int b0 = tid * CH, b1 = min(M, b0 + CH);
    double sum = 0, sq = 0;
    for (int i = b0; i < b1; ++i) { const double v = B.dist[i]; sum += v; sq += v * v; }
    red[0][tid] = sum; red[1][tid] = sq;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if (tid < off) { red[0][tid] += red[0][tid + off]; red[1][tid] += red[1][tid + off]; }
        __syncthreads();
    }
    if (tid == 0) {
        const double valid = (M >= MEAN_K + 1) ? (double)M : 0.0;
        const double mean = red[0][0] / valid;
        const double variance = (red[1][0] - red[0][0] * red[0][0] / valid) / (valid - 1);
        s_thr = mean + STD_MUL * sqrt(variance);
    }
    __syncthreads();
    const double thr = s_thr;
    int c = 0;
    for (int i = b0; i < b1; ++i) c += !((double)B.dist[i] > thr) ? 1 : 0;     // NaN threshold keeps everything
    int M2;
    int pos = block_excl_sum(c, lds, &M2);
    for (int i = b0; i < b1; ++i)
        if (!((double)B.dist[i] > thr)) {
            B.p2[pos] = B.p1[i];
            B.p2[B.cap + pos] = B.p1[B.cap + i];
            B.p2[2 * B.cap + pos] = B.p1[2 * B.cap + i];
            B.label[pos] = pos;
            B.cnt[pos] = 0;
            ++pos;
        }
    if (tid == 0) B.ctl->M2 = M2;
}

// ---- connected components: one round of in-place min-label propagation --------------------------
__global__ __launch_bounds__(256) void k3_propagate(Det3dBufs B, int round)
{
    __shared__ float tx[256], ty[256], tz[256];
    __shared__ int tl[256];
    const int M2 = B.ctl->M2;
    if ((int)(blockIdx.x * 256) >= M2) return;
    if (round > 0 && B.ctl->changed[round - 1] == 0) return;       // converged in an earlier round
    const int tid = threadIdx.x;
    const int i = blockIdx.x * 256 + tid;
    const bool live = i < M2;
    const float *X = B.p2, *Y = B.p2 + B.cap, *Z = B.p2 + 2 * B.cap;
    const float px = live ? X[i] : 0.f, py = live ? Y[i] : 0.f, pz = live ? Z[i] : 0.f;
    const int old = live ? B.label[i] : 0;
    int m = old;
    for (int j0 = 0; j0 < M2; j0 += 256) {
        const int j = j0 + tid;
        __syncthreads();
        if (j < M2) { tx[tid] = X[j]; ty[tid] = Y[j]; tz[tid] = Z[j]; tl[tid] = B.label[j]; }
        __syncthreads();
        const int jn = min(256, M2 - j0);
        if (live)
            for (int t = 0; t < jn; ++t)
                if (d2f(px, py, pz, tx[t], ty[t], tz[t]) < TOL2) m = min(m, tl[t]);
    }
    if (live) {
        for (int hop = 0; hop < 4; ++hop) m = min(m, B.label[m]);      // pointer jumping
        if (m < old) {
            atomicMin(&B.label[i], m);
            atomicMin(&B.label[old], m);                               // pull the old root down as well
            B.ctl->changed[round] = 1;
        }
    }
}

// ---- sizes, gate, order, centroids ------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k3_finish(Det3dBufs B, float sx, float sy, float cs, float sn, int max_centers)
{
    __shared__ int s_root[RDET_MAX_CENTERS], s_size[RDET_MAX_CENTERS], s_rank[RDET_MAX_CENTERS];
    __shared__ int s_n;
    const int tid = threadIdx.x;
    const int M2 = B.ctl->M2;
    if (tid == 0) {
        s_n = 0;
        if (B.ctl->changed[ROUNDS - 1]) B.ctl->err = RDET_ERR_CAPACITY;     // label propagation did not converge
    }
    __syncthreads();
    for (int i = tid; i < M2; i += 1024) {
        int r = B.label[i];
        while (B.label[r] != r) r = B.label[r];                            // final root = smallest index of the component
        B.label[i] = r;
    }
    __syncthreads();
    for (int i = tid; i < M2; i += 1024) atomicAdd(&B.cnt[B.label[i]], 1);
    __syncthreads();
    // accepted components, in ascending root (= first member) order
    if (tid == 0) {
        int n = 0, err = 0;
        for (int i = 0; i < M2; ++i) {
            const int c = B.cnt[i];
            if (B.label[i] == i && c >= MIN_SZ && c <= MAX_SZ) {            // :70-71
                if (n < RDET_MAX_CENTERS) { s_root[n] = i; s_size[n] = c; ++n; }
                else err = RDET_ERR_CAPACITY;
            }
        }
        if (n > max_centers) { err = RDET_ERR_BUFFER; n = 0; }
        s_n = n;
        if (err) B.ctl->err = err;
    }
    __syncthreads();
    const int n = s_n;
    if (tid < n) {          // rank: size descending, then first member index ascending (roots are already ascending)
        int rank = 0;
        for (int k = 0; k < n; ++k)
            if (s_size[k] > s_size[tid] || (s_size[k] == s_size[tid] && k < tid)) ++rank;
        s_rank[tid] = rank;
    }
    __syncthreads();
    if (tid < n) {
#pragma clang fp contract(off)
        const int root = s_root[tid];
        const float *X = B.p2, *Y = B.p2 + B.cap;
        float cx = 0.f, cy = 0.f;
        for (int i = root; i < M2; ++i)                                     // float32 sum in index order (:94)
            if (B.label[i] == root) { cx += X[i]; cy += Y[i]; }
        cx /= (float)s_size[tid]; cy /= (float)s_size[tid];
        const int r = s_rank[tid];
        B.ctl->centers[2 * r] = (cs * cx + (-sn) * cy) + sx;                // :96 Project2D(s2b).cast<float>() * p
        B.ctl->centers[2 * r + 1] = (sn * cx + cs * cy) + sy;
    }
    if (tid == 0) B.ctl->K = n;
}

}  // namespace

struct rdet3d {
    rdet3d_options opt;
    double s2b[3];
    int max_points, device;
    hipStream_t stream;
    float *d_xyzi, *d_p1, *d_p2, *d_dist;
    int *d_label, *d_cnt;
    Det3dCtl *d_ctl, *h_ctl;
    float *h_stage;
    std::string hip_error;
};

#define DET3_TRY(h, expr)                                                           \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            if (h) (h)->hip_error = std::string(#expr) + ": " + hipGetErrorString(e_); \
            return RDET_ERR_HIP;                                                    \
        }                                                                           \
    } while (0)

extern "C" {

int rdet3d_create(const rdet3d_options *opt, const double s2b[3], int max_points, int device, rdet3d_t **out)
{
    if (!opt || !s2b || !out || max_points < 1) return RDET_ERR_INVALID;
    *out = nullptr;
    rdet3d_t *h = new (std::nothrow) rdet3d();
    if (!h) return RDET_ERR_INVALID;
    h->opt = *opt;
    std::memcpy(h->s2b, s2b, sizeof(double) * 3);
    h->max_points = max_points;
    h->device = device;
    const size_t np = (size_t)max_points;
    int rc = [&]() -> int {
        DET3_TRY(h, hipSetDevice(device));
        DET3_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        DET3_TRY(h, hipMalloc(&h->d_xyzi, 16 * np));
        DET3_TRY(h, hipMalloc(&h->d_p1, 12 * np));
        DET3_TRY(h, hipMalloc(&h->d_p2, 12 * np));
        DET3_TRY(h, hipMalloc(&h->d_dist, 4 * np));
        DET3_TRY(h, hipMalloc(&h->d_label, 4 * np));
        DET3_TRY(h, hipMalloc(&h->d_cnt, 4 * np));
        DET3_TRY(h, hipMalloc(&h->d_ctl, sizeof(Det3dCtl)));
        DET3_TRY(h, hipHostMalloc(&h->h_ctl, sizeof(Det3dCtl)));
        DET3_TRY(h, hipHostMalloc(&h->h_stage, 16 * np));
        return RDET_OK;
    }();
    if (rc != RDET_OK) { std::fprintf(stderr, "rdet3d_create: %s\n", h->hip_error.c_str()); rdet3d_destroy(h); return rc; }
    *out = h;
    return RDET_OK;
}

void rdet3d_destroy(rdet3d_t *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void *ptrs[] = {h->d_xyzi, h->d_p1, h->d_p2, h->d_dist, h->d_label, h->d_cnt, h->d_ctl};
    for (void *p : ptrs) (void)hipFree(p);
    if (h->h_ctl) (void)hipHostFree(h->h_ctl);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int rdet3d_handle_cloud(rdet3d_t *h, double stamp, const float *xyzi, int N, float *centers_xy, int max_centers,
                        int *K, double *obs_time)
{
    if (!h || !K || N < 0 || (N > 0 && !xyzi) || max_centers < 0 || (max_centers > 0 && !centers_xy))
        return RDET_ERR_INVALID;
    *K = 0;
    if (obs_time) *obs_time = stamp;                                  // :16
    if (N == 0) return RDET_OK;
    if (N > h->max_points) return RDET_ERR_CAPACITY;
    DET3_TRY(h, hipSetDevice(h->device));
    std::memcpy(h->h_stage, xyzi, sizeof(float) * 4 * (size_t)N);
    DET3_TRY(h, hipMemcpyAsync(h->d_xyzi, h->h_stage, sizeof(float) * 4 * (size_t)N, hipMemcpyHostToDevice, h->stream));
    Det3dBufs B;
    B.xyzi = h->d_xyzi; B.p1 = h->d_p1; B.p2 = h->d_p2; B.dist = h->d_dist; B.label = h->d_label; B.cnt = h->d_cnt;
    B.ctl = h->d_ctl; B.cap = h->max_points;
    const int blocks = (N + 255) / 256;
    hipLaunchKernelGGL(k3_filter, dim3(1), dim3(1024), 0, h->stream, B, N, h->opt.intensity_min);
    hipLaunchKernelGGL(k3_knn, dim3(blocks), dim3(256), 0, h->stream, B);
    hipLaunchKernelGGL(k3_sor, dim3(1), dim3(1024), 0, h->stream, B);
    for (int r = 0; r < ROUNDS; ++r) hipLaunchKernelGGL(k3_propagate, dim3(blocks), dim3(256), 0, h->stream, B, r);
    const float sa = (float)h->s2b[2];
    hipLaunchKernelGGL(k3_finish, dim3(1), dim3(1024), 0, h->stream, B, (float)h->s2b[0], (float)h->s2b[1], cosf(sa),
                       sinf(sa), max_centers < RDET_MAX_CENTERS ? max_centers : RDET_MAX_CENTERS);
    DET3_TRY(h, hipMemcpyAsync(h->h_ctl, h->d_ctl, sizeof(Det3dCtl), hipMemcpyDeviceToHost, h->stream));
    DET3_TRY(h, hipStreamSynchronize(h->stream));
    if (h->h_ctl->err) return h->h_ctl->err;
    *K = h->h_ctl->K;
    if (*K > 0) std::memcpy(centers_xy, h->h_ctl->centers, sizeof(float) * 2 * (size_t)*K);
    return RDET_OK;
}

}  // extern "C"
