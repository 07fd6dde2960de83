// det3d.hip -- MI355X-native 3D (point cloud) reflector detector behind include/rdet.h.
//
// Replaces reflector_detect::PointCloudReflectorDetect::HandlePointCloud (reference
// src/reflector_detect/point_cloud/point_cloud_reflector_detect.cc:9-106), whose arithmetic
// is PCL 1.7's (un-vendored; the PCL semantics this file implements are spelled out in DESIGN.md section 4):
//
//   k3_filter_*   intensity > threshold, order-preserving compaction (a workgroup per 1024-point tile: counts, then
//                 ballot scan + write; coalesced 16-byte reads)                        (:31-39)
//   k3_knn        StatisticalOutlierRemoval part 1: per point the MeanK+1 = 31 smallest float32
//                 squared distances (the query first), brute force; lane = point, candidates through the
//                 scalar cache, the sorted list lives in registers (min/max insertion chain), eight waves
//                 share the candidates of 64 points and one admission bound          (:43-47)
//   k3_sor        part 2: mean / (n-1)-variance in FP64, threshold, second compaction
//   k3_cc_*       EuclideanClusterExtraction as connected components of the radius-0.2 m graph: smallest-neighbour
//                 pointers, a snapshot of the chain tops, then a lock-free union-find for the few adjacent pairs whose
//                 tops differ (roots are only ever hooked under smaller roots: the label is the smallest index) (:65-74)
//   k3_finish     component sizes, size gate [4,160], order (size desc, first index asc)
//   k3_centroids  float32 centroids in index order (one wave per component, spread over the CUs), Rigid2f to base_link (:77-97)
//
// No kd-tree: after the intensity gate a cloud holds 10^2..10^4 points, for which the
// all-pairs distance sweep is a coalesced, LDS-tiled, embarrassingly parallel kernel, while a
// tree build would be pointer-chasing.  Nothing waits on the host between stages: the point
// counts M, M2 stay on the device and every grid is sized for the capacity.
#include "../../include/rdet.h"
#include "host_visible.h"

#include <hip/hip_runtime.h>

#include <chrono>
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
    int croot[RDET_MAX_CENTERS], csize[RDET_MAX_CENTERS], crank[RDET_MAX_CENTERS];   // accepted components: k3_finish -> k3_centroids
};

// what the kernels hand back, in pinned host memory: every slot is ONE 16-byte system-scope store that carries the call's
// number, polled by the host (no D2H copy, no wait for the completion signal; same scheme as det2d.hip)
struct Det3dSlot { float x, y; int seq, pad; };
struct Det3dHead { int K, err, M2, seq; };
struct Det3dHostOut {
    Det3dHead head;
    Det3dSlot centers[RDET_MAX_CENTERS];
};
typedef unsigned d3_u32x4 __attribute__((ext_vector_type(4)));
__device__ static void d3_host_store16(void *p, unsigned a, unsigned b, unsigned c, unsigned d)
{
    const d3_u32x4 v = {a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

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
    Det3dHostOut *hout;   // pinned host memory (device view)
    int seq;              // this call's number
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

// ---- intensity filter + order-preserving compaction, one workgroup per 1024-point tile: the tiles' survivor counts
// first (k3_filter_count), then every tile adds up the counts in front of it and writes (k3_filter_write).  One
// workgroup walking all tiles with two barriers each took 45 us for 29 k points; coalesced 16-byte reads.
__global__ __launch_bounds__(1024) void k3_filter_count(Det3dBufs B, int N, double intensity_min)
{
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * 1024 + tid;
    const bool keep = i < N && (double)((const float4 *)B.xyzi)[i].w > intensity_min;      // :33
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    if (tid == 0) {
        int c = 0;
        for (int w = 0; w < 16; ++w) c += wsum[w];
        B.cnt[blockIdx.x] = c;                                                   // B.cnt is rebuilt by k3_sor for its own use
    }
}
__global__ __launch_bounds__(1024) void k3_filter_write(Det3dBufs B, int N, double intensity_min)
{
    __shared__ int wsum[16];
    __shared__ int base;
    const int tid = threadIdx.x;
    if (tid < 64) {                                                              // survivors in the tiles before this one
        int c = 0;
        for (int w = tid; w < (int)blockIdx.x; w += 64) c += B.cnt[w];
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
        if (tid == 0) base = c;
    }
    __syncthreads();
    const int i = blockIdx.x * 1024 + tid;
    const float4 cur = (i < N) ? ((const float4 *)B.xyzi)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool keep = i < N && (double)cur.w > intensity_min;
    const int pos = tile_compact_pos(keep, wsum, &base);
    if (keep) { B.p1[pos] = cur.x; B.p1[B.cap + pos] = cur.y; B.p1[2 * B.cap + pos] = cur.z; }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) { B.ctl->M = base; B.ctl->M2 = 0; B.ctl->K = 0; B.ctl->err = 0; }   // base now includes this tile
}

// ---- SOR part 1: mean distance to the MeanK nearest neighbours ---------------------------------
// One workgroup = 64 query points (lane = point) x KNN_WAVES waves, each wave sweeping its share of the candidate
// tiles (in order of index distance from the queries' own tile: scan order is spatially coherent, so the
// lists tighten early).  The MeanK+1 smallest squared distances live in REGISTERS as a sorted list; a
// candidate enters through a min/max chain that runs only when some lane of the wave needs it.  The partial
// lists are merged through LDS in a tree.  The multiset of the 31 smallest values is exact, so the
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
constexpr int KNN_WAVES = 8;        // waves sharing the candidates of 64 queries
constexpr int KNN_TILE = 128;       // candidates per tile (the unit dealt to the waves)

__global__ __launch_bounds__(64 * KNN_WAVES) void k3_knn(Det3dBufs B)
{
    __shared__ float part[KNN_WAVES / 2][KNN][64];   // hand-over buffers of the merge tree
    __shared__ float thr[KNN_WAVES][64];             // every wave's current 31st-smallest distance per query (see below)
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
    // The waves see disjoint candidates, so each one's list alone tightens four times slower than the true one would --
    // and the insertion chain runs for every candidate that ANY lane still admits.  But a wave's 31st-smallest value is an
    // upper bound of the final one whatever subset it has seen, so the waves publish theirs and admit only below the
    // smallest: no barrier needed, a stale (larger) bound is still a bound; the merged multiset stays exact.
    thr[wave][lane] = INFINITY;
    float tau = INFINITY;
    const int ntiles = (M + KNN_TILE - 1) / KNN_TILE, qt = q0 / KNN_TILE;
    // tiles by distance from the queries' tile: qt, qt+1, qt-1, qt+2, ...; wave w takes every 4th of them.
    // A candidate is the same for all 64 lanes: its coordinates come through the SCALAR cache (uniform
    // addresses -> s_load), eight at a time, and enter the VALU as SGPR operands: no LDS, no vector loads.
    int valid = 0;                                            // valid tiles so far: the v-th one goes to wave v % 4
    for (int k = 0; k < 2 * ntiles; ++k) {
        const int t = (k & 1) ? qt + (k + 1) / 2 : qt - k / 2;
        if (t < 0 || t >= ntiles) continue;
        if ((valid++ % KNN_WAVES) != wave) continue;
        const int j0 = KNN_TILE * t, jn = min(KNN_TILE, M - j0);
        float nx[8], ny[8], nz[8];                            // the next group of eight is loaded while this one is used
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int j = j0 + u; nx[u] = X[j]; ny[u] = Y[j]; nz[u] = Z[j]; }   // contiguous: ONE s_load_dwordx8 per array (a clamp per element would split it); reads past M stay inside the padded buffers
        for (int c0 = 0; c0 < jn; c0 += 8) {
            float cx[8], cy[8], cz[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { cx[u] = nx[u]; cy[u] = ny[u]; cz[u] = nz[u]; }
            if (c0 + 8 < jn) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int j = j0 + c0 + 8 + u; nx[u] = X[j]; ny[u] = Y[j]; nz[u] = Z[j]; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float d2 = d2f(px, py, pz, cx[u], cy[u], cz[u]);
                const bool need = (c0 + u < jn) && d2 < L[KNN - 1] && d2 < tau;
                if (__any(need)) knn_insert(L, need ? d2 : INFINITY);
            }
            if ((c0 & 31) == 24) {                              // every 32 candidates: publish / refresh the shared bound
                thr[wave][lane] = L[KNN - 1];
                float t = thr[0][lane];
#pragma unroll
                for (int w = 1; w < KNN_WAVES; ++w) t = fminf(t, thr[w][lane]);
                tau = t;
            }
        }
    }
    // merge tree: in every round the upper half of the remaining waves hands its list to the lower half
    for (int half = KNN_WAVES / 2; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
#pragma unroll
            for (int q = 0; q < KNN; ++q) part[wave - half][q][lane] = L[q];
        }
        __syncthreads();
        if (wave < half) {
            for (int q = 0; q < KNN; ++q) {                   // the partner's list is ascending: stop once it cannot improve
                const float v = part[wave][q][lane];
                const bool need = v < L[KNN - 1];
                if (!__any(need)) break;
                knn_insert(L, need ? v : INFINITY);
            }
        }
        __syncthreads();
    }
    if (wave == 0) {
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
// (the previous version needed up to 64 propagation launches).  The XCDs' L2 caches are not coherent with each other
// for plain loads, and a device-scope atomic load is a ~1.5 us round trip, so the finds read parent[] through the
// caches: every value parent[x] has ever held is an ancestor of x for good (hooks attach roots under smaller indices,
// halving only shortcuts upwards), hence a stale read can only return an ancestor that is no longer the root -- never a
// wrong one.  What must be exact is the hook itself: the CAS on the larger root goes to memory and fails when that
// node has stopped being a root, and only then the finds are repeated with device-scope loads (FRESH).
template <bool FRESH>
__device__ static inline int uf_load(const int *p)
{
    if (FRESH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *(const volatile int *)p;
}
template <bool FRESH>
__device__ static int uf_find(int *parent, int x)
{
    int p = uf_load<FRESH>(&parent[x]);
    while (p != x) {
        const int gp = uf_load<FRESH>(&parent[p]);
        if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // path halving
        x = p; p = gp;
    }
    return x;
}
// Three passes.  Doing every union inside the all-pairs sweep serialises: a candidate that is adjacent to ANY lane of the
// wave makes the whole wave walk through a dependent chain of memory operations (finds, CAS), ~100 such candidates per
// wave = 150 us.  Instead:
//   k3_cc_min   sweep 1, registers only: parent[i] = smallest index among i and its neighbours.  That alone puts nearly
//               every point of a compact cluster in one tree (the chains run towards the cluster's first point);
//   k3_cc_jump  root[i] = top of i's chain (a snapshot; plain loads);
//   k3_cc_link  sweep 2: the candidate's snapshot root arrives through the scalar cache with its coordinates, and only
//               an adjacent pair whose snapshot roots differ goes into the union code -- a few per cluster.
constexpr int CC_WAVES = 16;        // waves per 64 queries in the two sweeps: the sweep is a chain of scalar-cache misses per wave
__global__ __launch_bounds__(64 * CC_WAVES) void k3_cc_min(Det3dBufs B)
{
    __shared__ int s_min[CC_WAVES][64];
    const int M2 = B.ctl->M2;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int q0 = blockIdx.x * 64; q0 < M2; q0 += gridDim.x * 64) {      // the grid is capped: tiles of 64 queries, strided
    const int i = q0 + lane;
    const bool live = i < M2;
    const float *__restrict__ X = B.p2, *__restrict__ Y = B.p2 + B.cap, *__restrict__ Z = B.p2 + 2 * B.cap;
    const float px = live ? X[i] : 0.f, py = live ? Y[i] : 0.f, pz = live ? Z[i] : 0.f;
    int mi = i;
    const int jend = min(M2, q0 + 63);                        // only j < i
    float nx[8], ny[8], nz[8];                                // the next group of eight is loaded while this one is used
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int j = 8 * wave + u; nx[u] = X[j]; ny[u] = Y[j]; nz[u] = Z[j]; }   // contiguous -> s_load_dwordx8; past M2: padding, masked by j < i
    for (int c0 = 8 * wave; c0 < jend; c0 += 8 * CC_WAVES) {  // candidates through the scalar cache, ascending within a wave
        float cx[8], cy[8], cz[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { cx[u] = nx[u]; cy[u] = ny[u]; cz[u] = nz[u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int j = c0 + 8 * CC_WAVES + u; nx[u] = X[j]; ny[u] = Y[j]; nz[u] = Z[j]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = c0 + u;
            const bool adj = j < i && d2f(px, py, pz, cx[u], cy[u], cz[u]) < TOL2;
            mi = adj ? min(mi, j) : mi;
        }
        if (__ballot(live && mi == i) == 0ull) break;          // every lane has its (for this wave) smallest neighbour
    }
    s_min[wave][lane] = mi;
    __syncthreads();
    if (wave == 0 && live) {
        int m = s_min[0][lane];
#pragma unroll
        for (int w = 1; w < CC_WAVES; ++w) m = min(m, s_min[w][lane]);
        B.label[i] = m;
    }
    __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k3_cc_jump(Det3dBufs B)
{
    const int M2 = B.ctl->M2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M2) return;
    const int *parent = B.label;
    int r = parent[i];
    for (int p = parent[r]; p != r; p = parent[r]) r = p;     // parent[] is constant during this kernel
    reinterpret_cast<int *>(B.dist)[i] = r;                   // the SOR distances are dead: snapshot roots live there
}
__global__ __launch_bounds__(64 * CC_WAVES) void k3_cc_link(Det3dBufs B)
{
    const int M2 = B.ctl->M2;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int q0 = blockIdx.x * 64; q0 < M2; q0 += gridDim.x * 64) {
    const int i = q0 + lane;
    const bool live = i < M2;
    const float *__restrict__ X = B.p2, *__restrict__ Y = B.p2 + B.cap, *__restrict__ Z = B.p2 + 2 * B.cap;
    const int *__restrict__ root = reinterpret_cast<const int *>(B.dist);
    const float px = live ? X[i] : 0.f, py = live ? Y[i] : 0.f, pz = live ? Z[i] : 0.f;
    int *parent = B.label;
    const int rs = live ? root[i] : -1;                       // snapshot root of i
    int ri = rs;                                              // a (possibly stale) ancestor of i
    int rm = rs;                                              // snapshot root of the tree merged last
    const int jend = min(M2, q0 + 63);                        // only j < i: every edge is handled by its larger end
    float nx[8], ny[8], nz[8];
    int nr[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int j = 8 * wave + u; nx[u] = X[j]; ny[u] = Y[j]; nz[u] = Z[j]; nr[u] = root[j]; }
    for (int c0 = 8 * wave; c0 < jend; c0 += 8 * CC_WAVES) {
        float cx[8], cy[8], cz[8];
        int cr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { cx[u] = nx[u]; cy[u] = ny[u]; cz[u] = nz[u]; cr[u] = nr[u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int j = c0 + 8 * CC_WAVES + u; nx[u] = X[j]; ny[u] = Y[j]; nz[u] = Z[j]; nr[u] = root[j]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = c0 + u;
            if (live && j < i && cr[u] != rs && cr[u] != rm && d2f(px, py, pz, cx[u], cy[u], cz[u]) < TOL2) {
                int a = uf_find<false>(parent, ri), b = uf_find<false>(parent, cr[u]);
                while (a != b) {
                    const int hi = max(a, b), lo = min(a, b);
                    const int old = atomicCAS(&parent[hi], hi, lo);
                    if (old == hi) { a = lo; break; }
                    a = uf_find<true>(parent, a); b = uf_find<true>(parent, b);
                }
                ri = a;
                rm = cr[u];
            }
        }
    }
    }
}

// ---- sizes, gate, order, centroids ------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k3_finish(Det3dBufs B, int max_centers)
{
    __shared__ int s_root[RDET_MAX_CENTERS], s_size[RDET_MAX_CENTERS], s_rank[RDET_MAX_CENTERS];
    __shared__ int wsum[16];
    __shared__ int base;
    __shared__ int s_err;
    const int tid = threadIdx.x;
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
    if (tid < n) { B.ctl->croot[tid] = s_root[tid]; B.ctl->csize[tid] = s_size[tid]; B.ctl->crank[tid] = s_rank[tid]; }
    if (tid == 0) {
        B.ctl->K = n;
        d3_host_store16(&B.hout->head, (unsigned)n, (unsigned)s_err, (unsigned)M2, (unsigned)B.seq);   // the centres follow, each with its own tag
    }
}

// Centroids: one WAVE per accepted component, four per workgroup, spread over the CUs (inside the single workgroup of
// k3_finish sixteen waves took turns on up to 256 components).  Members are found 64 at a time; the float32 sums run in
// index order (:94).
__global__ __launch_bounds__(256) void k3_centroids(Det3dBufs B, float sx, float sy, float cs, float sn)
{
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cidx = blockIdx.x * 4 + wave;
    if (cidx >= B.ctl->K) return;
    const float *X = B.p2, *Y = B.p2 + B.cap;
    const int root = B.ctl->croot[cidx], last = B.last[root];
    float cx = 0.f, cy = 0.f;
    // a component's members sit in one stripe per scan ring, far apart in index: most 64-point chunks hold none, so
    // four chunks' labels are fetched per round trip and only the chunks with members pay for the coordinates
    for (int b0 = root; b0 <= last; b0 += 256) {
        int lab[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = b0 + 64 * u + lane; lab[u] = (i <= last) ? B.label[i] : -1; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = b0 + 64 * u + lane;
            const bool mem = lab[u] == root;
            unsigned long long mask = __ballot(mem);
            if (mask == 0ull) continue;
            const float x = mem ? X[i] : 0.f, y = mem ? Y[i] : 0.f;
            while (mask) {
                const int b = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                cx += __shfl(x, b, 64);
                cy += __shfl(y, b, 64);
            }
        }
    }
    if (lane == 0) {
        const float sz = (float)B.ctl->csize[cidx];
        cx /= sz; cy /= sz;
        const int r = B.ctl->crank[cidx];
        const float ox = (cs * cx + (-sn) * cy) + sx, oy = (sn * cx + cs * cy) + sy;   // :96 Project2D(s2b).cast<float>() * p
        B.ctl->centers[2 * r] = ox;
        B.ctl->centers[2 * r + 1] = oy;
        d3_host_store16(&B.hout->centers[r], __float_as_uint(ox), __float_as_uint(oy), (unsigned)B.seq, 0u);
    }
}

}  // namespace

struct rdet3d {
    rdet3d_options opt;
    double s2b[3];
    int max_points, device;
    hipStream_t stream;
    float *d_xyzi, *d_p1, *d_p2, *d_dist;
    int *d_label, *d_cnt, *d_last;
    Det3dCtl *d_ctl;
    Det3dHostOut *h_out, *dv_out;      // pinned + mapped: polled result slots (host / device view)
    bool xyzi_in_vram;                 // d_xyzi is fine-grained device memory the host writes through the PCIe BAR (else: pinned staging + copy)
    float *h_stage;
    int seq;
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
        h->d_xyzi = (float *)host_visible::alloc(16 * np);
        if (h->d_xyzi) h->xyzi_in_vram = true;
        else {
            h->xyzi_in_vram = false;
            DET3_TRY(h, hipMalloc(&h->d_xyzi, 16 * np));
            DET3_TRY(h, hipHostMalloc(&h->h_stage, 16 * np));
        }
        // + 1024 floats: the sweeps read candidates eight at a time through the scalar cache, unclamped, up to a few groups past the end
        DET3_TRY(h, hipMalloc(&h->d_p1, 12 * np + 4096)); DET3_TRY(h, hipMemset(h->d_p1, 0, 12 * np + 4096));
        DET3_TRY(h, hipMalloc(&h->d_p2, 12 * np + 4096)); DET3_TRY(h, hipMemset(h->d_p2, 0, 12 * np + 4096));
        DET3_TRY(h, hipMalloc(&h->d_dist, 4 * np + 4096)); DET3_TRY(h, hipMemset(h->d_dist, 0, 4 * np + 4096));
        DET3_TRY(h, hipMalloc(&h->d_label, 4 * np));
        DET3_TRY(h, hipMalloc(&h->d_cnt, 4 * np));
        DET3_TRY(h, hipMalloc(&h->d_last, 4 * np));
        DET3_TRY(h, hipMalloc(&h->d_ctl, sizeof(Det3dCtl)));
        DET3_TRY(h, hipHostMalloc(&h->h_out, sizeof(Det3dHostOut), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(h->h_out, 0, sizeof(Det3dHostOut));
        void *dv = nullptr;
        DET3_TRY(h, hipHostGetDevicePointer(&dv, h->h_out, 0)); h->dv_out = (Det3dHostOut *)dv;
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
    if (h->h_out) (void)hipHostFree(h->h_out);
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
    DET3_TRY(h, hipStreamSynchronize(h->stream));                      // (the previous call returned on its last result slot, not on the kernels' end)
    if (h->xyzi_in_vram) {                                             // the cloud goes straight into device memory: posted writes, no copy engine
        std::memcpy(h->d_xyzi, xyzi, sizeof(float) * 4 * (size_t)N);
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
    } else {
        std::memcpy(h->h_stage, xyzi, sizeof(float) * 4 * (size_t)N);
        DET3_TRY(h, hipMemcpyAsync(h->d_xyzi, h->h_stage, sizeof(float) * 4 * (size_t)N, hipMemcpyHostToDevice, h->stream));
    }
    Det3dBufs B;
    B.xyzi = h->d_xyzi; B.p1 = h->d_p1; B.p2 = h->d_p2; B.dist = h->d_dist; B.label = h->d_label; B.cnt = h->d_cnt; B.last = h->d_last;
    B.ctl = h->d_ctl; B.cap = h->max_points;
    B.hout = h->dv_out; B.seq = ++h->seq;
    const int blocks = (N + 63) / 64;                                 // 64 query points per workgroup; M <= N stays on the device
    const int ftiles = N > 0 ? (N + 1023) / 1024 : 1;
    hipLaunchKernelGGL(k3_filter_count, dim3(ftiles), dim3(1024), 0, h->stream, B, N, h->opt.intensity_min);
    hipLaunchKernelGGL(k3_filter_write, dim3(ftiles), dim3(1024), 0, h->stream, B, N, h->opt.intensity_min);
    hipLaunchKernelGGL(k3_knn, dim3(blocks), dim3(64 * KNN_WAVES), 0, h->stream, B);
    hipLaunchKernelGGL(k3_sor, dim3(1), dim3(1024), 0, h->stream, B);
    const int cc_blocks = blocks < 256 ? blocks : 256;                 // grid-stride over the query tiles: M2 is only known on the device
    hipLaunchKernelGGL(k3_cc_min, dim3(cc_blocks), dim3(64 * CC_WAVES), 0, h->stream, B);
    hipLaunchKernelGGL(k3_cc_jump, dim3((h->max_points + 255) / 256), dim3(256), 0, h->stream, B);
    hipLaunchKernelGGL(k3_cc_link, dim3(cc_blocks), dim3(64 * CC_WAVES), 0, h->stream, B);
    const float sa = (float)h->s2b[2];
    hipLaunchKernelGGL(k3_finish, dim3(1), dim3(1024), 0, h->stream, B, max_centers < RDET_MAX_CENTERS ? max_centers : RDET_MAX_CENTERS);
    hipLaunchKernelGGL(k3_centroids, dim3(RDET_MAX_CENTERS / 4), dim3(256), 0, h->stream, B, (float)h->s2b[0], (float)h->s2b[1], cosf(sa), sinf(sa));
    DET3_TRY(h, hipGetLastError());
    // poll the head (written by k3_finish), then each centre's own tag (k3_centroids)
    auto wait_tag = [&](const int *tag) -> int {
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (__atomic_load_n(tag, __ATOMIC_ACQUIRE) != B.seq) {
            if ((++spins & 0xfffffu) == 0) {
                if (hipStreamQuery(h->stream) != hipErrorNotReady) {
                    DET3_TRY(h, hipStreamSynchronize(h->stream));
                    if (__atomic_load_n(tag, __ATOMIC_ACQUIRE) == B.seq) break;
                    h->hip_error = "the 3D detector's kernels finished without publishing their result";
                    return RDET_ERR_HIP;
                }
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) { h->hip_error = "rdet3d: no result after 10 s"; return RDET_ERR_HIP; }
            }
        }
        return RDET_OK;
    };
    int rc = wait_tag(&h->h_out->head.seq);
    if (rc != RDET_OK) return rc;
    const Det3dHead head = h->h_out->head;
    if (head.err) return head.err;
    *K = head.K;
    for (int c = 0; c < head.K; ++c) {
        rc = wait_tag(&h->h_out->centers[c].seq);
        if (rc != RDET_OK) return rc;
        centers_xy[2 * c] = h->h_out->centers[c].x; centers_xy[2 * c + 1] = h->h_out->centers[c].y;
    }
    return RDET_OK;
}

}  // extern "C"
