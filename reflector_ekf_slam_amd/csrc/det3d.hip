// det3d.hip -- MI355X-native 3D (point cloud) reflector detector behind include/rdet.h.
//
// Replaces reflector_detect::PointCloudReflectorDetect::HandlePointCloud (reference
// src/reflector_detect/point_cloud/point_cloud_reflector_detect.cc:9-106), whose arithmetic
// is PCL 1.7's (un-vendored; the PCL semantics this file implements are spelled out in DESIGN.md section 4):
//
//   k3_filter     intensity > threshold, order-preserving compaction (ballot scan per 1024-point
//                 tile, coalesced 16-byte reads)                                   (:31-39)
//   k3_knn        StatisticalOutlierRemoval part 1: per point the MeanK+1 = 31 smallest float32
//                 squared distances (the query first), brute force over LDS-staged candidate
//                 tiles; lane = point, the sorted list lives in registers (min/max insertion
//                 chain), four waves share the candidates of 64 points                  (:43-47)
//   k3_sor        part 2: mean / (n-1)-variance in FP64, threshold, second compaction
//   k3_cc         EuclideanClusterExtraction as connected components of the radius-0.2 m graph:
//                 one all-pairs pass with a lock-free union-find (roots are only ever hooked under
//                 smaller roots, so a component's label is its smallest index)        (:65-74)
//   k3_finish     component sizes, size gate [4,160], order (size desc, first index asc),
//                 float32 centroids in index order (one wave per component), Rigid2f to base_link (:77-97)
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

struct Det3dCtl {
    int M, M2, K, err;
    float centers[2 * RDET_MAX_CENTERS];
};

struct Det3dBufs {
    const float *xyzi;
    float *p1;        // 3 x cap, SoA: x | y | z  after the intensity filter
    float *p2;        // after SOR
    float *dist;      // SOR mean neighbour distance
    int *label;
    int *cnt;
    int *last;        // last member index per root
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

// order-preserving block compaction step for one tile of 1024 candidates: returns this thread's output
// position (valid when flag) and adds the tile's count to *base (all threads see the new value afterwards)
__device__ static int tile_compact_pos(bool flag, int *wsum, int *base)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long bal = __ballot(flag);
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int off = *base, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int c = wsum[w]; if (w < wave) off += c; tot += c; }
    __syncthreads();
    if (tid == 0) *base += tot;
    __syncthreads();
    return off + __popcll(bal & lt);
}

// ---- intensity filter + compaction (one workgroup keeps the point order; coalesced 16-byte reads) --
__global__ __launch_bounds__(1024) void k3_filter(Det3dBufs B, int N, double intensity_min)
{
    __shared__ int wsum[16];
    __shared__ int base;
    const int tid = threadIdx.x;
    if (tid == 0) base = 0;
    __syncthreads();
    const float4 *src = (const float4 *)B.xyzi;
    float4 p = (tid < N) ? src[tid] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t0 = 0; t0 < N; t0 += 1024) {
        const int i = t0 + tid;
        const float4 cur = p;
        if (i + 1024 < N) p = src[i + 1024];                                      // next tile in flight
        const bool keep = i < N && (double)cur.w > intensity_min;                // :33
        const int pos = tile_compact_pos(keep, wsum, &base);
        if (keep) { B.p1[pos] = cur.x; B.p1[B.cap + pos] = cur.y; B.p1[2 * B.cap + pos] = cur.z; }
    }
    if (tid == 0) { B.ctl->M = base; B.ctl->M2 = 0; B.ctl->K = 0; B.ctl->err = 0; }
}

// ---- SOR part 1: mean distance to the MeanK nearest neighbours ---------------------------------
// One workgroup = 64 query points (lane = point) x 4 waves, each wave sweeping a quarter of the candidate
// tiles (in order of index distance from the queries' own tile: scan order is spatially coherent, so the
// lists tighten early).  The MeanK+1 smallest squared distances live in REGISTERS as a sorted list; a
// candidate enters through a min/max chain that runs only when some lane of the wave needs it.  The four
// partial lists are merged through LDS by wave 0.  The multiset of the 31 smallest values is exact, so the
// ascending-order FP64 sum below is bit-identical to the insertion-sort reference.
constexpr int KNN = MEAN_K + 1;
__device__ static inline void knn_insert(float (&L)[KNN], float x)
{
#pragma unroll
    for (int q = 0; q < KNN; ++q) {
        const float lo = fminf(L[q], x);
        x = fmaxf(L[q], x);
        L[q] = lo;
    }
}
__global__ __launch_bounds__(256) void k3_knn(Det3dBufs B)
{
    __shared__ float part[3][KNN][64];            // partial lists of waves 1..3
    const int M = B.ctl->M;
    const int q0 = blockIdx.x * 64;
    if (q0 >= M) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = q0 + lane;
    const bool live = i < M;
    const float *__restrict__ X = B.p1, *__restrict__ Y = B.p1 + B.cap, *__restrict__ Z = B.p1 + 2 * B.cap;
    const float px = live ? X[i] : 0.f, py = live ? Y[i] : 0.f, pz = live ? Z[i] : 0.f;
    float L[KNN];
#pragma unroll
    for (int q = 0; q < KNN; ++q) L[q] = INFINITY;
    const int ntiles = (M + 255) / 256, qt = q0 / 256;
    // tiles by distance from the queries' tile: qt, qt+1, qt-1, qt+2, ...; wave w takes every 4th of them.
    // A candidate is the same for all 64 lanes: its coordinates come through the SCALAR cache (uniform
    // addresses -> s_load), eight at a time, and enter the VALU as SGPR operands: no LDS, no vector loads.
    int valid = 0;                                            // valid tiles so far: the v-th one goes to wave v % 4
    for (int k = 0; k < 2 * ntiles; ++k) {
        const int t = (k & 1) ? qt + (k + 1) / 2 : qt - k / 2;
        if (t < 0 || t >= ntiles) continue;
        if ((valid++ & 3) != wave) continue;
        const int j0 = 256 * t, jn = min(256, M - j0);
        float nx[8], ny[8], nz[8];                            // the next group of eight is loaded while this one is used
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int j = min(j0 + u, M - 1); nx[u] = X[j]; ny[u] = Y[j]; nz[u] = Z[j]; }
        for (int c0 = 0; c0 < jn; c0 += 8) {
            float cx[8], cy[8], cz[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { cx[u] = nx[u]; cy[u] = ny[u]; cz[u] = nz[u]; }
            if (c0 + 8 < jn) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int j = min(j0 + c0 + 8 + u, M - 1); nx[u] = X[j]; ny[u] = Y[j]; nz[u] = Z[j]; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float d2 = d2f(px, py, pz, cx[u], cy[u], cz[u]);
                const bool need = (c0 + u < jn) && d2 < L[KNN - 1];
                if (__any(need)) knn_insert(L, need ? d2 : INFINITY);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int q = 0; q < KNN; ++q) part[wave - 1][q][lane] = L[q];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
            for (int q = 0; q < KNN; ++q) {                   // each partial list is ascending: stop once it cannot improve
                const float v = part[w][q][lane];
                const bool need = v < L[KNN - 1];
                if (!__any(need)) break;
                knn_insert(L, need ? v : INFINITY);
            }
        if (live) {
            float dst = 0.f;                                      // search "failed": fewer than MeanK+1 points
            if (M >= KNN) {
                double dist_sum = 0;
#pragma unroll
                for (int k = 1; k < KNN; ++k) dist_sum += sqrtf(L[k]);   // k = 0 is the query itself
                dst = (float)(dist_sum / MEAN_K);
            }
            B.dist[i] = dst;
        }
    }
}

// ---- SOR part 2: statistics, threshold, second compaction ---------------------------------------
__global__ __launch_bounds__(1024) void k3_sor(Det3dBufs B)
{
    __shared__ int wsum[16];
    __shared__ int base;
    __shared__ double red[2][1024];
    __shared__ double s_thr;
    const int tid = threadIdx.x;
    const int M = B.ctl->M;
    const int CH = (M + 1023) / 1024;
    const int b0 = tid * CH, b1 = min(M, b0 + CH);
    double sum = 0, sq = 0;
    for (int i = b0; i < b1; ++i) { const double v = B.dist[i]; sum += v; sq += v * v; }   // chunked like the serial loop's partial sums
    red[0][tid] = sum; red[1][tid] = sq;
    if (tid == 0) base = 0;
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
    for (int t0 = 0; t0 < M; t0 += 1024) {
        const int i = t0 + tid;
        const bool keep = i < M && !((double)B.dist[i] > thr);                  // NaN threshold keeps everything
        const int pos = tile_compact_pos(keep, wsum, &base);
        if (keep) {
            B.p2[pos] = B.p1[i];
            B.p2[B.cap + pos] = B.p1[B.cap + i];
            B.p2[2 * B.cap + pos] = B.p1[2 * B.cap + i];
            B.label[pos] = pos;
            B.cnt[pos] = 0;
            B.last[pos] = 0;
        }
    }
    if (tid == 0) B.ctl->M2 = base;
}

// ---- connected components of the radius graph: lock-free union-find ------------------------------
// parent = B.label.  Only roots are ever hooked (CAS root -> a SMALLER root), so the final root of a
// component is its smallest index whatever the interleaving: deterministic labels from one all-pairs pass
// (the previous version needed up to 64 propagation launches).  All accesses to parent[] are device-scope
// atomics: the XCDs' L2 caches are not coherent with each other for plain loads/stores.
__device__ static inline int uf_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ static int uf_find(int *parent, int x)
{
    int p = uf_load(&parent[x]);
    while (p != x) {
        const int gp = uf_load(&parent[p]);
        if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // path halving
        x = p; p = gp;
    }
    return x;
}
__global__ __launch_bounds__(256) void k3_cc(Det3dBufs B)
{
    const int M2 = B.ctl->M2;
    const int q0 = blockIdx.x * 64;
    if (q0 >= M2) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = q0 + lane;
    const bool live = i < M2;
    const float *__restrict__ X = B.p2, *__restrict__ Y = B.p2 + B.cap, *__restrict__ Z = B.p2 + 2 * B.cap;
    const float px = live ? X[i] : 0.f, py = live ? Y[i] : 0.f, pz = live ? Z[i] : 0.f;
    int *parent = B.label;
    int ri = i;                                               // a (possibly stale) ancestor of i
    const int jend = min(M2, q0 + 63);                        // only j < i: every edge is handled by its larger end
    // candidates through the scalar cache as in k3_knn; wave w takes the 8-candidate groups w, w+4, ...
    for (int c0 = 8 * wave; c0 < jend; c0 += 32) {
        float cx[8], cy[8], cz[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = min(c0 + u, M2 - 1);
            cx[u] = X[j]; cy[u] = Y[j]; cz[u] = Z[j];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = c0 + u;
            if (live && j < i && d2f(px, py, pz, cx[u], cy[u], cz[u]) < TOL2) {
                int a = uf_find(parent, ri), b = uf_find(parent, j);
                while (a != b) {
                    const int hi = max(a, b), lo = min(a, b);
                    const int old = atomicCAS(&parent[hi], hi, lo);
                    if (old == hi) { a = lo; break; }
                    a = uf_find(parent, a); b = uf_find(parent, b);
                }
                ri = a;
            }
        }
    }
}

// ---- sizes, gate, order, centroids ------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k3_finish(Det3dBufs B, float sx, float sy, float cs, float sn, int max_centers)
{
    __shared__ int s_root[RDET_MAX_CENTERS], s_size[RDET_MAX_CENTERS], s_rank[RDET_MAX_CENTERS];
    __shared__ int wsum[16];
    __shared__ int base;
    __shared__ int s_err;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M2 = B.ctl->M2;
    if (tid == 0) { base = 0; s_err = 0; }
    __syncthreads();
    for (int i = tid; i < M2; i += 1024) {
        int r = B.label[i];
        while (B.label[r] != r) r = B.label[r];                            // final root = smallest index of the component
        B.label[i] = r;
    }
    __syncthreads();
    for (int i = tid; i < M2; i += 1024) { const int r = B.label[i]; atomicAdd(&B.cnt[r], 1); atomicMax(&B.last[r], i); }
    __syncthreads();
    // accepted components, in ascending root (= first member) order
    for (int t0 = 0; t0 < M2; t0 += 1024) {
        const int i = t0 + tid;
        const int c = (i < M2) ? B.cnt[i] : 0;
        const bool ok = i < M2 && B.label[i] == i && c >= MIN_SZ && c <= MAX_SZ;   // :70-71
        const int pos = tile_compact_pos(ok, wsum, &base);
        if (ok) {
            if (pos < RDET_MAX_CENTERS) { s_root[pos] = i; s_size[pos] = c; }
            else s_err = RDET_ERR_CAPACITY;
        }
    }
    __syncthreads();
    int n = min(base, RDET_MAX_CENTERS);
    if (n > max_centers) { if (tid == 0) s_err = RDET_ERR_BUFFER; n = 0; }
    __syncthreads();
    if (tid == 0 && s_err) B.ctl->err = s_err;
    if (tid < n) {          // rank: size descending, then first member index ascending (roots are already ascending)
        int rank = 0;
        for (int k = 0; k < n; ++k)
            if (s_size[k] > s_size[tid] || (s_size[k] == s_size[tid] && k < tid)) ++rank;
        s_rank[tid] = rank;
    }
    __syncthreads();
    // centroids: one wave per component; members are found 64 at a time, the float32 sums run in index order (:94)
    const float *X = B.p2, *Y = B.p2 + B.cap;
    for (int cidx = wave; cidx < n; cidx += 16) {
#pragma clang fp contract(off)
        const int root = s_root[cidx], last = B.last[root];
        float cx = 0.f, cy = 0.f;
        for (int b0 = root; b0 <= last; b0 += 64) {
            const int i = b0 + lane;
            const bool mem = i <= last && B.label[i] == root;
            const float x = mem ? X[i] : 0.f, y = mem ? Y[i] : 0.f;
            unsigned long long mask = __ballot(mem);
            while (mask) {
                const int b = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                cx += __shfl(x, b, 64);
                cy += __shfl(y, b, 64);
            }
        }
        if (lane == 0) {
            cx /= (float)s_size[cidx]; cy /= (float)s_size[cidx];
            const int r = s_rank[cidx];
            B.ctl->centers[2 * r] = (cs * cx + (-sn) * cy) + sx;                // :96 Project2D(s2b).cast<float>() * p
            B.ctl->centers[2 * r + 1] = (sn * cx + cs * cy) + sy;
        }
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
    int *d_label, *d_cnt, *d_last;
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
        DET3_TRY(h, hipMalloc(&h->d_last, 4 * np));
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
    void *ptrs[] = {h->d_xyzi, h->d_p1, h->d_p2, h->d_dist, h->d_label, h->d_cnt, h->d_last, h->d_ctl};
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
    B.xyzi = h->d_xyzi; B.p1 = h->d_p1; B.p2 = h->d_p2; B.dist = h->d_dist; B.label = h->d_label; B.cnt = h->d_cnt; B.last = h->d_last;
    B.ctl = h->d_ctl; B.cap = h->max_points;
    const int blocks = (N + 63) / 64;                                 // 64 query points per workgroup; M <= N stays on the device
    hipLaunchKernelGGL(k3_filter, dim3(1), dim3(1024), 0, h->stream, B, N, h->opt.intensity_min);
    hipLaunchKernelGGL(k3_knn, dim3(blocks), dim3(256), 0, h->stream, B);
    hipLaunchKernelGGL(k3_sor, dim3(1), dim3(1024), 0, h->stream, B);
    hipLaunchKernelGGL(k3_cc, dim3(blocks), dim3(256), 0, h->stream, B);
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
