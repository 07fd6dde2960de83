// det3d.hip -- MI355X-native 3D (point cloud) reflector detector behind include/rdet.h.
//
// Replaces reflector_detect::PointCloudReflectorDetect::HandlePointCloud (reference
// src/reflector_detect/point_cloud/point_cloud_reflector_detect.cc:9-106), whose arithmetic
// is PCL 1.7's (un-vendored; the PCL semantics this file implements are spelled out in DESIGN.md section 4):
//
//   k3_filter_*   intensity > threshold, order-preserving compaction (a workgroup per 1024-point tile: counts, then
//                 ballot scan + write; coalesced 16-byte reads); beside them the grid-cell histogram of the survivors
//                 and its scan                                                          (:31-39)
//   k3_scatter,   the survivors in Morton order of a 128 x 128 grid over (x, y) + the bounding box of every 32
//   k3_boxes      consecutive ones: what lets the two neighbour searches below skip almost everything
//   k3_knn        StatisticalOutlierRemoval part 1: per point the MeanK+1 = 31 smallest float32
//                 squared distances (the query first); lane = point, candidates through the
//                 scalar cache, the sorted list lives in registers (min/max insertion chain), eight waves
//                 share the candidate tiles of 64 points and one admission bound; a tile whose box is farther
//                 than every lane's bound is skipped                                   (:43-47)
//   k3_sor        part 2: mean / (n-1)-variance in FP64, threshold; outliers masked in the sorted copy
//   k3_cc_*       EuclideanClusterExtraction as connected components of the radius-0.2 m graph: smallest-neighbour
//                 pointers, a snapshot of the chain tops, then a lock-free union-find for the few adjacent pairs whose
//                 tops differ (roots are only ever hooked under smaller roots: the label is the smallest index) (:65-74)
//   k3_finish_a/b component sizes (over the CUs), size gate [4,160], order (size desc, first index asc)
//   k3_centroids  float32 centroids in index order (one wave per component, spread over the CUs), Rigid2f to base_link (:77-97)
//
// No kd-tree: after the intensity gate a cloud holds 10^2..10^4 points; a counting sort into Morton order and a box per
// 32 points prune as well as a tree would at this size and stay coalesced, data-parallel and free of pointer chasing.
// Nothing waits on the host between stages: the point counts M, M2 stay on the device and every grid is sized for the
// number of points that came in.
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

// The spatial order (round 4).  Both neighbour searches of the reference -- the 31 nearest neighbours of
// StatisticalOutlierRemoval and the 0.2 m radius graph of EuclideanClusterExtraction -- are local, but the cloud arrives
// ring by ring: one post is spread over all rings, so in arrival order every 64 points needed every other point.  The
// survivors of the intensity gate are therefore brought into MORTON ORDER of a GRID_G x GRID_G grid over (x, y) (one
// counting sort: histogram beside the gate, scan beside the compaction, scatter), every BOX_PTS consecutive points get
// their bounding box, and the sweeps -- unchanged otherwise: lane = query, candidates through the scalar cache -- first
// test a candidate tile's box against each lane's own bound and skip it when no lane can use it.  Nothing of the result
// depends on the order: the k-NN multiset is exact (a tile is skipped only if its box is farther than the lane's current
// 31st distance, with a margin far above the float32 rounding of both sides), node ids stay the ARRIVAL indices (labels =
// smallest arrival index, centroids summed in arrival order), so neither the grid's placement nor the order of the
// points inside a cell (atomic cursors: not deterministic) is visible in the output.
constexpr int GRID_G = 128, GRID_CELLS = GRID_G * GRID_G;
constexpr int BOX_PTS = 32;         // points per bounding box = candidates per tile
constexpr float BOX_MARGIN = 0.9999f;   // box distance^2 * margin < bound  <=>  "some point of the box may matter"

struct Det3dCtl {
    int M, M2, K, err;
    float centers[2 * RDET_MAX_CENTERS];
    int croot[RDET_MAX_CENTERS], csize[RDET_MAX_CENTERS], crank[RDET_MAX_CENTERS];   // accepted components: k3_finish_b -> k3_centroids
    // the grid the NEXT cloud is sorted on = this cloud's survivor bounding box (clouds of one sensor look alike; only
    // the sweeps' pruning, never a result, depends on it).  bb = this cloud's box, ordered-int encoded, by atomics.
    int bb[4];
    float gx0, gy0, ginv;
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
    float *p1;        // 3 x cap, SoA: x | y | z  after the intensity filter, arrival order ("node" numbering)
    float *s1;        // the same points in Morton order; k3_sor overwrites the outliers' x with NaN
    int *perm;        // sorted position -> node
    float *box;       // 8 floats per BOX_PTS sorted points: min x, y, z, max x, y, z
    int *hist;        // GRID_CELLS cell counts (zero between calls)
    int *cursor;      // GRID_CELLS scatter cursors
    float *dist;      // per node: SOR mean neighbour distance; later, per sorted position: snapshot roots
    int *label;       // per node: union-find parent, -1 = removed by SOR
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

// squared distance from a point to a box (a lower bound of d2f to every point inside it, up to rounding: BOX_MARGIN)
__device__ static inline float box_d2(float px, float py, float pz, float x0, float y0, float z0, float x1, float y1, float z1)
{
    const float dx = fmaxf(fmaxf(x0 - px, px - x1), 0.f);
    const float dy = fmaxf(fmaxf(y0 - py, py - y1), 0.f);
    const float dz = fmaxf(fmaxf(z0 - pz, pz - z1), 0.f);
    return dx * dx + dy * dy + dz * dz;
}

__device__ static inline int enc_ord(float f) { const int b = __float_as_int(f); return b >= 0 ? b : b ^ 0x7fffffff; }
__device__ static inline float dec_ord(int e) { return __int_as_float(e >= 0 ? e : e ^ 0x7fffffff); }

// Morton code of the grid cell of (x, y); non-finite and out-of-grid coordinates are clamped to the border cells (the
// clamp is monotone, which is all the sort has to be: locality is a matter of speed, never of the result)
__device__ static inline int cell_code(float x, float y, float gx0, float gy0, float ginv)
{
    const float fx = fminf(fmaxf((x - gx0) * ginv, 0.f), (float)(GRID_G - 1));
    const float fy = fminf(fmaxf((y - gy0) * ginv, 0.f), (float)(GRID_G - 1));
    unsigned cx = (unsigned)(int)fx, cy = (unsigned)(int)fy;
    cx = (cx | (cx << 4)) & 0x0f0fu; cx = (cx | (cx << 2)) & 0x3333u; cx = (cx | (cx << 1)) & 0x5555u;
    cy = (cy | (cy << 4)) & 0x0f0fu; cy = (cy | (cy << 2)) & 0x3333u; cy = (cy | (cy << 1)) & 0x5555u;
    return (int)(cx | (cy << 1));
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
// Round 4: the survivors' grid-cell histogram and bounding box are taken beside the count, the cell scan beside the write.
__global__ __launch_bounds__(1024) void k3_filter_count(Det3dBufs B, int N, double intensity_min)
{
    __shared__ int wsum[16];
    __shared__ int s_bb[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * 1024 + tid;
    if (tid < 4) s_bb[tid] = (tid < 2) ? 0x7fffffff : (int)0x80000000;
    const float4 cur = (i < N) ? ((const float4 *)B.xyzi)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool keep = i < N && (double)cur.w > intensity_min;      // :33
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    if (keep) {
        atomicAdd(&B.hist[cell_code(cur.x, cur.y, B.ctl->gx0, B.ctl->gy0, B.ctl->ginv)], 1);
        if (fabsf(cur.x) < 1e30f && fabsf(cur.y) < 1e30f) {
            const int ex = enc_ord(cur.x), ey = enc_ord(cur.y);
            atomicMin(&s_bb[0], ex); atomicMin(&s_bb[1], ey); atomicMax(&s_bb[2], ex); atomicMax(&s_bb[3], ey);
        }
    }
    __syncthreads();
    if (tid == 0) {
        int c = 0;
        for (int w = 0; w < 16; ++w) c += wsum[w];
        B.cnt[blockIdx.x] = c;                                                   // B.cnt is rebuilt by k3_scatter for its own use
    }
    if (tid < 2 && s_bb[tid] != 0x7fffffff) atomicMin(&B.ctl->bb[tid], s_bb[tid]);
    else if (tid >= 2 && tid < 4 && s_bb[tid] != (int)0x80000000) atomicMax(&B.ctl->bb[tid], s_bb[tid]);
}
// workgroups [0, ftiles): the compaction; workgroups [ftiles, ftiles + GRID_CELLS / 1024): the exclusive scan of the cell
// histogram into the scatter cursors, 1024 cells each (a chunk adds up the chunks in front of it by itself)
__global__ __launch_bounds__(1024) void k3_filter_write(Det3dBufs B, int N, double intensity_min, int ftiles)
{
    __shared__ int wsum[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x >= ftiles) {
        __shared__ int wpre[16];
        const int c = blockIdx.x - ftiles;
        int pre = 0;
        for (int k = tid; k < c * 1024; k += 1024) pre += B.hist[k];
        const int v = B.hist[c * 1024 + tid];
        int incl = v;                                                             // inclusive scan over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
        for (int off = 32; off > 0; off >>= 1) pre += __shfl_xor(pre, off, 64);
        if (lane == 63) wsum[wave] = incl;
        if (lane == 0) wpre[wave] = pre;
        __syncthreads();
        int off = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { off += wpre[w]; if (w < wave) off += wsum[w]; }
        B.cursor[c * 1024 + tid] = off + incl - v;
        return;
    }
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
    if ((int)blockIdx.x == ftiles - 1 && tid == 0) { B.ctl->M = base; B.ctl->M2 = 0; B.ctl->K = 0; B.ctl->err = 0; }   // base now includes this tile
}

// bounding boxes of BOX_PTS consecutive sorted points: lane = point, 32-lane halves reduce by shuffles.  NaN coordinates
// (the outliers k3_sor masks) are ignored by fminf / fmaxf; a tile without any number gets an empty box (+inf, -inf).
__device__ static inline void tile_boxes(const Det3dBufs &B, int s, int M)
{
    const float *X = B.s1, *Y = B.s1 + B.cap, *Z = B.s1 + 2 * B.cap;
    const bool v = s < M;
    const float qn = __int_as_float(0x7fc00000);
    const float x = v ? X[s] : qn, y = v ? Y[s] : qn, z = v ? Z[s] : qn;
    const bool ok = x == x;                                                       // x carries the mask
    float m[6] = {ok ? x : qn, ok ? y : qn, ok ? z : qn, ok ? x : qn, ok ? y : qn, ok ? z : qn};
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { m[k] = fminf(m[k], __shfl_xor(m[k], off, 64)); m[3 + k] = fmaxf(m[3 + k], __shfl_xor(m[3 + k], off, 64)); }
    }
    if ((threadIdx.x & 31) == 0 && (s & ~31) < M) {
        float *b = B.box + 8 * (s >> 5);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            b[k] = (m[k] == m[k]) ? m[k] : INFINITY;
            b[3 + k] = (m[3 + k] == m[3 + k]) ? m[3 + k] : -INFINITY;
        }
    }
}

// ---- the counting sort's scatter (thread = node): sorted coordinates + the permutation; also clears what the next
// stages and the next call expect cleared
__global__ __launch_bounds__(1024) void k3_scatter(Det3dBufs B)
{
    const int M = B.ctl->M;
    const int gid = blockIdx.x * 1024 + threadIdx.x;
    for (int k = gid; k < GRID_CELLS; k += gridDim.x * 1024) B.hist[k] = 0;
    if (gid >= M) return;
    const float x = B.p1[gid], y = B.p1[B.cap + gid], z = B.p1[2 * B.cap + gid];
    const int pos = atomicAdd(&B.cursor[cell_code(x, y, B.ctl->gx0, B.ctl->gy0, B.ctl->ginv)], 1);
    B.s1[pos] = x; B.s1[B.cap + pos] = y; B.s1[2 * B.cap + pos] = z;
    B.perm[pos] = gid;
    B.cnt[gid] = 0; B.last[gid] = 0;
}
__global__ __launch_bounds__(256) void k3_boxes(Det3dBufs B)
{
    const int M = B.ctl->M;
    if ((int)blockIdx.x * 256 >= M) return;
    tile_boxes(B, blockIdx.x * 256 + threadIdx.x, M);
}

// ---- SOR part 1: mean distance to the MeanK nearest neighbours ---------------------------------
// One workgroup = 64 query points, consecutive in the spatial order (lane = point) x KNN_WAVES waves, each wave taking
// its share of the candidate tiles in order of index distance from the queries' own tile (near in Morton order is
// near in space, so the lists tighten at once) and skipping every tile whose box no lane can use.  The MeanK+1
// smallest squared distances live in REGISTERS as a sorted list; a candidate enters through a min/max chain that runs
// only when some lane of the wave needs it.  The partial lists are merged through LDS in a tree.  The multiset of the 31
// smallest values is exact, so the ascending-order FP64 sum below is bit-identical to the insertion-sort reference.
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

__global__ __launch_bounds__(64 * KNN_WAVES) void k3_knn(Det3dBufs B)
{
    __shared__ float part[KNN_WAVES / 2][KNN][64];   // hand-over buffers of the merge tree
    __shared__ float thr[KNN_WAVES][64];             // every wave's current 31st-smallest distance per query (see below)
    const int M = B.ctl->M;
    const int q0 = blockIdx.x * 64;
    if (q0 >= M) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = q0 + lane;
    const bool live = s < M;
    const float *__restrict__ X = B.s1, *__restrict__ Y = B.s1 + B.cap, *__restrict__ Z = B.s1 + 2 * B.cap;
    const float px = live ? X[s] : 0.f, py = live ? Y[s] : 0.f, pz = live ? Z[s] : 0.f;
    float L[KNN];
#pragma unroll
    for (int q = 0; q < KNN; ++q) L[q] = INFINITY;
    // The waves see disjoint candidates, so each one's list alone tightens slower than the true one would -- and the
    // insertion chain runs for every candidate that ANY lane still admits.  But a wave's 31st-smallest value is an
    // upper bound of the final one whatever subset it has seen, so the waves publish theirs and admit only below the
    // smallest: no barrier needed, a stale (larger) bound is still a bound; the merged multiset stays exact.
    thr[wave][lane] = INFINITY;
    float tau = INFINITY;
    // tiles by distance from the queries' tile: qt, qt+1, qt-1, qt+2, ... (one-sided once a border is reached); wave w
    // takes the w-th, (w + KNN_WAVES)-th ... of them.  A candidate is the same for all 64 lanes: its box and its
    // coordinates come through the SCALAR cache (uniform addresses -> s_load) and enter the VALU as SGPR operands.
    const int ntiles = (M + BOX_PTS - 1) / BOX_PTS, qt = q0 / BOX_PTS;
    const int na = qt, nb = ntiles - 1 - qt, nm = min(na, nb);
    for (int v = wave; v < ntiles; v += KNN_WAVES) {
        int t;
        if (v <= 2 * nm) t = (v & 1) ? qt + (v + 1) / 2 : qt - v / 2;
        else t = (nb > na) ? qt + (v - nm) : qt - (v - nm);
        const float *__restrict__ bx = B.box + 8 * t;
        const float db = box_d2(px, py, pz, bx[0], bx[1], bx[2], bx[3], bx[4], bx[5]);
        if (!__any(live && db * BOX_MARGIN < fminf(L[KNN - 1], tau))) continue;
        const int j0 = BOX_PTS * t, jn = min(BOX_PTS, M - j0);
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
                const bool need = live && (c0 + u < jn) && d2 < L[KNN - 1] && d2 < tau;
                if (__any(need)) knn_insert(L, need ? d2 : INFINITY);
            }
        }
        thr[wave][lane] = L[KNN - 1];                            // every tile: publish / refresh the shared bound
        float tm = thr[0][lane];
#pragma unroll
        for (int w = 1; w < KNN_WAVES; ++w) tm = fminf(tm, thr[w][lane]);
        tau = tm;
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
            B.dist[B.perm[s]] = dst;
        }
    }
}

// ---- SOR part 2: statistics, threshold; the outliers are masked in the sorted copy (x := NaN: every distance to them
// compares false), the union-find parents are initialised (removed node: -1), the boxes are rebuilt without them.  Node
// ids stay the arrival indices after the intensity gate -- the reference renumbers the survivors, but only the ORDER of
// the indices is ever used (smallest member, centroid summation), and the renumbering keeps the order.
__global__ __launch_bounds__(1024) void k3_sor(Det3dBufs B)
{
    __shared__ double red[2][1024];
    __shared__ double s_thr;
    __shared__ int s_m2;
    const int tid = threadIdx.x;
    const int M = B.ctl->M;
    const int CH = (M + 1023) / 1024;
    const int b0 = tid * CH, b1 = min(M, b0 + CH);
    double sum = 0, sq = 0;
    for (int i = b0; i < b1; ++i) { const double v = B.dist[i]; sum += v; sq += v * v; }   // chunked like the serial loop's partial sums
    red[0][tid] = sum; red[1][tid] = sq;
    if (tid == 0) s_m2 = 0;
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
    int kept = 0;
    for (int s = tid; s < M; s += 1024) {
        const int node = B.perm[s];
        const bool keep = !((double)B.dist[node] > thr);                        // NaN threshold keeps everything
        B.label[node] = keep ? node : -1;
        if (!keep) B.s1[s] = __int_as_float(0x7fc00000);
        kept += keep;
    }
    for (int off = 32; off > 0; off >>= 1) kept += __shfl_xor(kept, off, 64);
    if ((tid & 63) == 0) atomicAdd(&s_m2, kept);
    __syncthreads();                                                              // (also: the masked s1 is visible to the workgroup)
    if (tid == 0) B.ctl->M2 = s_m2;
    for (int s0 = 0; s0 < M; s0 += 1024) tile_boxes(B, s0 + tid, M);
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
// Three passes.  Doing every union inside the sweep serialises: a candidate that is adjacent to ANY lane of the
// wave makes the whole wave walk through a dependent chain of memory operations (finds, CAS), ~100 such candidates per
// wave = 150 us.  Instead:
//   k3_cc_min   sweep 1, registers only: parent[i] = smallest index among i and its neighbours.  That alone puts nearly
//               every point of a compact cluster in one tree (the chains run towards the cluster's first point);
//   k3_cc_jump  root[i] = top of i's chain (a snapshot; plain loads);
//   k3_cc_link  sweep 2: the candidate's snapshot root arrives through the scalar cache with its coordinates, and only
//               an adjacent pair whose snapshot roots differ goes into the union code -- a few per cluster.
// Both sweeps run over the sorted copy and skip every candidate tile whose box is farther than 0.2 m from all 64 queries.
constexpr int CC_WAVES = 8;         // waves per 64 queries in the two sweeps
__global__ __launch_bounds__(64 * CC_WAVES) void k3_cc_min(Det3dBufs B)
{
    __shared__ int s_min[CC_WAVES][64];
    const int M = B.ctl->M;
    const int ntiles = (M + BOX_PTS - 1) / BOX_PTS;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *__restrict__ X = B.s1, *__restrict__ Y = B.s1 + B.cap, *__restrict__ Z = B.s1 + 2 * B.cap;
    const int *__restrict__ P = B.perm;
    const float qn = __int_as_float(0x7fc00000);
    for (int q0 = blockIdx.x * 64; q0 < M; q0 += gridDim.x * 64) {        // the grid is capped: tiles of 64 queries, strided
        const int s = q0 + lane;
        const bool live = s < M;
        const float px = live ? X[s] : qn, py = live ? Y[s] : 0.f, pz = live ? Z[s] : 0.f;   // a masked / dead query is adjacent to nothing
        const int own = live ? P[s] : 0x7fffffff;
        int mi = own;
        for (int t = wave; t < ntiles; t += CC_WAVES) {
            const float *__restrict__ bx = B.box + 8 * t;
            if (!__any(box_d2(px, py, pz, bx[0], bx[1], bx[2], bx[3], bx[4], bx[5]) * BOX_MARGIN < TOL2)) continue;
            const int j0 = BOX_PTS * t, jn = min(BOX_PTS, M - j0);
#pragma unroll 1
            for (int c0 = 0; c0 < BOX_PTS; c0 += 8) {
                float cx[8], cy[8], cz[8];
                int ci[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int j = j0 + c0 + u; cx[u] = X[j]; cy[u] = Y[j]; cz[u] = Z[j]; ci[u] = P[j]; }   // contiguous -> s_load_dwordx8; past M: padding, masked below
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool adj = (c0 + u < jn) && d2f(px, py, pz, cx[u], cy[u], cz[u]) < TOL2;
                    mi = adj ? min(mi, ci[u]) : mi;
                }
            }
        }
        s_min[wave][lane] = mi;
        __syncthreads();
        if (wave == 0 && live && px == px) {
            int m = s_min[0][lane];
#pragma unroll
            for (int w = 1; w < CC_WAVES; ++w) m = min(m, s_min[w][lane]);
            B.label[own] = m;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k3_cc_jump(Det3dBufs B)
{
    const int M = B.ctl->M;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= M) return;
    const int *parent = B.label;
    int r = parent[B.perm[s]];
    if (r >= 0)
        for (int p = parent[r]; p != r; p = parent[r]) r = p;     // parent[] is constant during this kernel
    reinterpret_cast<int *>(B.dist)[s] = r;                       // the SOR distances are dead: snapshot roots live there, by sorted position
}
__global__ __launch_bounds__(64 * CC_WAVES) void k3_cc_link(Det3dBufs B)
{
    const int M = B.ctl->M;
    const int ntiles = (M + BOX_PTS - 1) / BOX_PTS;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *__restrict__ X = B.s1, *__restrict__ Y = B.s1 + B.cap, *__restrict__ Z = B.s1 + 2 * B.cap;
    const int *__restrict__ root = reinterpret_cast<const int *>(B.dist);
    const float qn = __int_as_float(0x7fc00000);
    int *parent = B.label;
    for (int q0 = blockIdx.x * 64; q0 < M; q0 += gridDim.x * 64) {
        const int s = q0 + lane;
        const bool live = s < M;
        const float px = live ? X[s] : qn, py = live ? Y[s] : 0.f, pz = live ? Z[s] : 0.f;
        const int rs = live ? root[s] : -1;                       // snapshot root of the query
        int ri = rs;                                              // a (possibly stale) ancestor of it
        int rm = rs;                                              // snapshot root of the tree merged last
        for (int t = wave; t < ntiles; t += CC_WAVES) {
            const float *__restrict__ bx = B.box + 8 * t;
            if (!__any(box_d2(px, py, pz, bx[0], bx[1], bx[2], bx[3], bx[4], bx[5]) * BOX_MARGIN < TOL2)) continue;
            const int j0 = BOX_PTS * t;
#pragma unroll 1
            for (int c0 = 0; c0 < BOX_PTS; c0 += 8) {
                float cx[8], cy[8], cz[8];
                int cr[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int j = j0 + c0 + u; cx[u] = X[j]; cy[u] = Y[j]; cz[u] = Z[j]; cr[u] = root[j]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + c0 + u;                    // every edge is handled by its later end (j < s also keeps j < M)
                    if (rs >= 0 && j < s && cr[u] >= 0 && cr[u] != rs && cr[u] != rm && d2f(px, py, pz, cx[u], cy[u], cz[u]) < TOL2) {
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
}

// ---- sizes, gate, order, centroids ------------------------------------------------------------------
// k3_finish_a (thread = node, over the CUs): final roots, component sizes and last members -- one atomic pair per
// (wave, component) instead of one per point: consecutive arrival indices mostly share their component.
__global__ __launch_bounds__(256) void k3_finish_a(Det3dBufs B)
{
    const int M = B.ctl->M;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if ((int)blockIdx.x * 256 >= M) return;
    int r = -1;
    if (i < M) {
        r = B.label[i];
        if (r >= 0) {
            while (true) { const int p = *(const volatile int *)&B.label[r]; if (p == r) break; r = p; }   // parents are final: no union runs any more
            B.label[i] = r;                                                   // (a root's own entry is left alone; others only move up)
        }
    }
    unsigned long long todo = __ballot(r >= 0);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        const int r0 = __shfl(r, src, 64);
        const unsigned long long grp = __ballot(r == r0);
        if ((int)(threadIdx.x & 63) == src) {
            atomicAdd(&B.cnt[r0], __popcll(grp));
            atomicMax(&B.last[r0], (int)(blockIdx.x * 256 + (threadIdx.x & ~63u)) + 63 - __clzll((long long)grp));
        }
        todo &= ~grp;
    }
}
__global__ __launch_bounds__(1024) void k3_finish_b(Det3dBufs B, int max_centers)
{
    __shared__ int s_root[RDET_MAX_CENTERS], s_size[RDET_MAX_CENTERS], s_rank[RDET_MAX_CENTERS];
    __shared__ int wsum[16];
    __shared__ int base;
    __shared__ int s_err;
    const int tid = threadIdx.x;
    const int M = B.ctl->M;
    if (tid == 0) { base = 0; s_err = 0; }
    __syncthreads();
    // accepted components, in ascending root (= first member) order
    for (int t0 = 0; t0 < M; t0 += 1024) {
        const int i = t0 + tid;
        const int c = (i < M) ? B.cnt[i] : 0;
        const bool ok = i < M && c >= MIN_SZ && c <= MAX_SZ && B.label[i] == i;   // :70-71 (cnt is non-zero at roots only)
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
        d3_host_store16(&B.hout->head, (unsigned)n, (unsigned)s_err, (unsigned)B.ctl->M2, (unsigned)B.seq);   // the centres follow, each with its own tag
        // the next cloud's grid: this cloud's survivor box, a little wider, at least 1/8 m per cell
        if (B.ctl->bb[0] != 0x7fffffff) {
            const float x0 = dec_ord(B.ctl->bb[0]), y0 = dec_ord(B.ctl->bb[1]), x1 = dec_ord(B.ctl->bb[2]), y1 = dec_ord(B.ctl->bb[3]);
            const float ext = fmaxf(fmaxf(x1 - x0, y1 - y0) * 1.05f, 0.125f * GRID_G);
            B.ctl->gx0 = 0.5f * (x0 + x1) - 0.5f * ext; B.ctl->gy0 = 0.5f * (y0 + y1) - 0.5f * ext; B.ctl->ginv = (float)GRID_G / ext;
        }
        B.ctl->bb[0] = B.ctl->bb[1] = 0x7fffffff; B.ctl->bb[2] = B.ctl->bb[3] = (int)0x80000000;
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
    const float *X = B.p1, *Y = B.p1 + B.cap;
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
    float *d_xyzi, *d_p1, *d_s1, *d_dist, *d_box;
    int *d_label, *d_cnt, *d_last, *d_perm, *d_hist, *d_cursor;
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
        DET3_TRY(h, hipMalloc(&h->d_s1, 12 * np + 4096)); DET3_TRY(h, hipMemset(h->d_s1, 0, 12 * np + 4096));
        DET3_TRY(h, hipMalloc(&h->d_perm, 4 * np + 4096)); DET3_TRY(h, hipMemset(h->d_perm, 0, 4 * np + 4096));
        DET3_TRY(h, hipMalloc(&h->d_box, 32 * (np / BOX_PTS + 2)));
        DET3_TRY(h, hipMalloc(&h->d_hist, 4 * GRID_CELLS)); DET3_TRY(h, hipMemset(h->d_hist, 0, 4 * GRID_CELLS));
        DET3_TRY(h, hipMalloc(&h->d_cursor, 4 * GRID_CELLS));
        DET3_TRY(h, hipMalloc(&h->d_dist, 4 * np + 4096)); DET3_TRY(h, hipMemset(h->d_dist, 0, 4 * np + 4096));
        DET3_TRY(h, hipMalloc(&h->d_label, 4 * np));
        DET3_TRY(h, hipMalloc(&h->d_cnt, 4 * np));
        DET3_TRY(h, hipMalloc(&h->d_last, 4 * np));
        DET3_TRY(h, hipMalloc(&h->d_ctl, sizeof(Det3dCtl)));
        {   // the first cloud is sorted on a 64 m x 64 m grid around the sensor; every later one on its predecessor's box
            Det3dCtl c0;
            std::memset(&c0, 0, sizeof(c0));
            c0.bb[0] = c0.bb[1] = 0x7fffffff; c0.bb[2] = c0.bb[3] = (int)0x80000000;
            c0.gx0 = c0.gy0 = -32.f; c0.ginv = (float)GRID_G / 64.f;
            DET3_TRY(h, hipMemcpy(h->d_ctl, &c0, sizeof(c0), hipMemcpyHostToDevice));
        }
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
    void *ptrs[] = {h->d_xyzi, h->d_p1, h->d_s1, h->d_dist, h->d_box, h->d_label, h->d_cnt, h->d_last, h->d_perm, h->d_hist, h->d_cursor, h->d_ctl};
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
    B.xyzi = h->d_xyzi; B.p1 = h->d_p1; B.s1 = h->d_s1; B.perm = h->d_perm; B.box = h->d_box; B.hist = h->d_hist; B.cursor = h->d_cursor;
    B.dist = h->d_dist; B.label = h->d_label; B.cnt = h->d_cnt; B.last = h->d_last;
    B.ctl = h->d_ctl; B.cap = h->max_points;
    B.hout = h->dv_out; B.seq = ++h->seq;
    const int blocks = (N + 63) / 64;                                 // 64 query points per workgroup; M <= N stays on the device
    const int ftiles = (N + 1023) / 1024, b256 = (N + 255) / 256;
    hipLaunchKernelGGL(k3_filter_count, dim3(ftiles), dim3(1024), 0, h->stream, B, N, h->opt.intensity_min);
    hipLaunchKernelGGL(k3_filter_write, dim3(ftiles + GRID_CELLS / 1024), dim3(1024), 0, h->stream, B, N, h->opt.intensity_min, ftiles);
    hipLaunchKernelGGL(k3_scatter, dim3(ftiles), dim3(1024), 0, h->stream, B);
    hipLaunchKernelGGL(k3_boxes, dim3(b256), dim3(256), 0, h->stream, B);
    hipLaunchKernelGGL(k3_knn, dim3(blocks), dim3(64 * KNN_WAVES), 0, h->stream, B);
    hipLaunchKernelGGL(k3_sor, dim3(1), dim3(1024), 0, h->stream, B);
    const int cc_blocks = blocks < 1024 ? blocks : 1024;               // grid-stride over the query tiles: M is only known on the device
    hipLaunchKernelGGL(k3_cc_min, dim3(cc_blocks), dim3(64 * CC_WAVES), 0, h->stream, B);
    hipLaunchKernelGGL(k3_cc_jump, dim3(b256), dim3(256), 0, h->stream, B);
    hipLaunchKernelGGL(k3_cc_link, dim3(cc_blocks), dim3(64 * CC_WAVES), 0, h->stream, B);
    const float sa = (float)h->s2b[2];
    hipLaunchKernelGGL(k3_finish_a, dim3(b256), dim3(256), 0, h->stream, B);
    hipLaunchKernelGGL(k3_finish_b, dim3(1), dim3(1024), 0, h->stream, B, max_centers < RDET_MAX_CENTERS ? max_centers : RDET_MAX_CENTERS);
    hipLaunchKernelGGL(k3_centroids, dim3(RDET_MAX_CENTERS / 4), dim3(256), 0, h->stream, B, (float)h->s2b[0], (float)h->s2b[1], cosf(sa), sinf(sa));
    DET3_TRY(h, hipGetLastError());
    // poll the head (written by k3_finish_b), then each centre's own tag (k3_centroids)
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
