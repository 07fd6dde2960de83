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
//   k3_knn        StatisticalOutlierRemoval part 1: per point the MeanK+1 = 31 smallest float32 squared distances (the
//                 query first).  ONE WAVE PER QUERY, lane = candidate: the 64 smallest distances so far live one per lane
//                 (wave-level bitonic networks on DPP / permlane swaps), tiles are opened nearest first and only while
//                 their box is nearer than the query's current 31st distance                (:43-47)
//   k3_cc_min,    part 2 -- mean / (n-1)-variance in FP64, threshold: every workgroup for itself -- and EuclideanClusterExtraction as
//   k3_cc_link    connected components of the radius-0.2 m graph, one wave per query over the tiles whose box lies within 0.2 m:
//                 smallest-neighbour pointers (outliers: out of the graph, and out of the boxes), then a lock-free union-find
//                 for the few adjacent pairs whose chain tops differ (roots are only ever hooked under smaller roots: the
//                 label is the smallest arrival index)                                    (:43-47, :65-74)
//   k3_finish_a   final roots, component sizes and extents in the sorted copy, the list of roots (over the CUs)
//   k3_clusters   size gate [4,160], order (size desc, first index asc), float32 centroids in index order (one wave
//                 per component, spread over the CUs), Rigid2f to base_link             (:70-71, :77-97)
//
// Two chains (round 6).  A cloud whose survivors of the gate number at most MFAST = 5120 (judged by the previous cloud's count) takes FIVE
// launches: k3f_front (gate + sort + boxes: every tile compacts its survivors in place, and the tile workgroup that arrives last -- no
// workgroup ever waits for another -- sorts them in LDS), k3_knn, k3_cc_min, k3_cc_link, k3f_clusters (final roots, sizes, gate, order and
// centroids from an LDS copy of the forest).  Anything bigger takes the NINE launches listed above; a cloud that turns out too big for the
// short front end is sent again through the long chain by the collecting call (D3_RETRY).  Nothing of the result depends on the chain
// (tests/test_detect3d_paths_gpu.py forces every cloud through both).
// Every launch is a short chain of dependent trips to memory.  Round 5 tried FIVE launches another way -- gate + sort + boxes as one launch
// whose scatter workgroups WAIT for its tile workgroups, k-NN with the statistics by its last workgroup, finish + clusters likewise
// -- bit-identical, and SLOWER (120 us against 88): every edge of this chain is all-to-all, and a hand-over inside a launch that
// crosses the XCDs' L2s (device-scope stores, a counter, a poll, device-scope loads; or release / acquire fences, which the L2
// serialises at 13 ns a wave) costs 3 - 5 us where a kernel boundary costs 1.5 - 2 (profiles/r05_chain_experiments.txt, item 12).  What
// pays is a hand-over nobody waits at, to ONE workgroup (a few dozen arrivals: 1 us), for work that fits one CU's LDS
// (profiles/r06_experiments.txt, items 14 - 17); and what is inside the kernels: first loads asked for together instead of one behind the
// other, the union-find's reads through the caches, one union per distinct neighbour tree, the statistics inside k3_cc_min.
// No kd-tree: after the intensity gate a cloud holds 10^2..10^4 points; a counting sort into Morton order and a box per
// 32 points prune as well as a tree would at this size and stay coalesced, data-parallel and free of pointer chasing.
// Nothing waits on the host between stages: the point count M stays on the device and every grid is sized for the
// number of points that came in.
#include "../../include/rdet.h"
#include "host_visible.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
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
// their bounding box, and the sweeps test a candidate tile's box against the query's own bound before they open it.  Nothing
// of the result depends on the order: the k-NN multiset is exact (a tile is skipped only if its box is farther than the query's
// current 31st distance, with a margin far above the float32 rounding of both sides), node ids stay the ARRIVAL indices (labels =
// smallest arrival index, centroids summed in arrival order), so neither the grid's placement nor the order of the
// points inside a cell (atomic cursors: not deterministic) is visible in the output.
constexpr int GRID_G = 128, GRID_CELLS = GRID_G * GRID_G;
constexpr int BOX_PTS = 32;         // points per bounding box = candidates per tile
constexpr float BOX_MARGIN = 0.9999f;   // box distance^2 * margin < bound  <=>  "some point of the box may matter"

struct Det3dCtl {
    int M, M2, K, err;
    int nroots;                     // length of Det3dBufs::roots (k3_finish)
    // the grid the NEXT cloud is sorted on = the bounding box of this cloud's inliers (clouds of one sensor look alike;
    // only the sweeps' pruning, never a result, depends on it): k3_finish
    float gx0, gy0, ginv;
    // the short front end (k3f_sort) found more survivors than it holds: every later kernel of this chain sees M = 0, the last one reports
    // D3_RETRY and the host sends the cloud through the long chain (m_true: the count, for the next cloud's choice)
    int retry, m_true;
    int arrived;                    // k3f_front: tile workgroups that have written their survivors (back to 0 when the launch ends)
};
constexpr int D3_RETRY = 1 << 20;   // (never leaves this file)

// what the kernels hand back, in pinned host memory: every slot is ONE 16-byte system-scope store that carries the call's
// number, polled by the host (no D2H copy, no wait for the completion signal; same scheme as det2d.hip)
struct Det3dSlot { float x, y; int seq, pad; };
struct Det3dHead { int K, err, M, seq; };      // (M: the survivors of the intensity gate -- the next call's hint for how many queries to expect)
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
    float *s1;        // the same points in Morton order
    int *perm;        // sorted position -> node (= arrival index among the survivors: the reference's point index)
    float *box;       // 8 floats per BOX_PTS sorted points: min x, y, z, max x, y, z -- of all survivors of the gate (k3_boxes)
    float *box2;      // the same without SOR's outliers (k3_cc_min writes, k3_cc_link and k3_clusters read)
    int *hist;        // GRID_CELLS cell counts (zero between calls)
    int *cursor;      // GRID_CELLS scatter cursors
    float *dist;      // per node: SOR mean neighbour distance
    float *dist_s;    // the same per sorted position; later the final root (k3_finish)
    int *label;       // per node: union-find parent, -1 = removed by SOR
    int *cnt;         // per root: component size
    int *first, *last;   // per root: first / last member's sorted position
    int *roots;       // all roots, in no particular order
    Det3dCtl *ctl;
    int cap;
    int cap1;         // p1's plane stride in the short front end: cap rounded up to whole 1024-point tiles
    Det3dHostOut *hout;   // pinned host memory (device view)
    int seq;              // this call's number
};

#ifdef RDET_DEBUG_MARKS
// in-kernel timelines (k3_clusters, k3_cc_link): wall_clock64() (100 MHz) per workgroup and phase; scripts/gpu_dbg_det3d.py
// (one table per kernel: 0 k3_clusters, 1 k3_cc_link, 2 k3_knn, 3 k3_cc_min, 4 k3f_sort, 5 k3f_clusters)
__device__ unsigned long long d3_marks_all[6][2048][8];
#define d3_marks d3_marks_all[D3_KERNEL]
#define D3_MARK(i) do { if (threadIdx.x == 0 && blockIdx.x < 2048) d3_marks[blockIdx.x][i] = wall_clock64(); } while (0)
#define D3_NOTE(i, v) do { if (threadIdx.x == 0 && blockIdx.x < 2048) d3_marks[blockIdx.x][i] = (unsigned long long)(v); } while (0)
#else
#define D3_MARK(i) do { } while (0)
#define D3_NOTE(i, v) do { } while (0)
#endif

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


// Morton code of the grid cell of (x, y); non-finite and out-of-grid coordinates are clamped to the border cells (the
// clamp is monotone, which is all the sort has to be: locality is a matter of speed, never of the result)
// (G: the grid's side; ginv is always GRID_G cells per extent, a coarser grid scales it)
template <int G = GRID_G>
__device__ static inline int cell_code(float x, float y, float gx0, float gy0, float ginv)
{
    const float gs = ginv * ((float)G / (float)GRID_G);
    const float fx = fminf(fmaxf((x - gx0) * gs, 0.f), (float)(G - 1));
    const float fy = fminf(fmaxf((y - gy0) * gs, 0.f), (float)(G - 1));
    unsigned cx = (unsigned)(int)fx, cy = (unsigned)(int)fy;
    cx = (cx | (cx << 4)) & 0x0f0fu; cx = (cx | (cx << 2)) & 0x3333u; cx = (cx | (cx << 1)) & 0x5555u;
    cy = (cy | (cy << 4)) & 0x0f0fu; cy = (cy | (cy << 2)) & 0x3333u; cy = (cy | (cy << 1)) & 0x5555u;
    return (int)(cx | (cy << 1));
}

// lane ^ J exchanges without the LDS crossbar (scripts/probe/lane_xor.hip checks them): DPP quad permutes, row shifts
// under bank masks, gfx950's v_permlane16_swap / v_permlane32_swap
template <int CTRL, int BANK>
__device__ static inline int d3_dpp(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, BANK, false); }
template <int J>
__device__ static inline float lane_xor(float f, int lane)
{
    const int v = __float_as_int(f);
    int r;
    if (J == 1) r = __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);             // (every lane has a source: nothing of the old value to keep)
    else if (J == 2) r = __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);
    else if (J == 4) r = d3_dpp<0x114, 0xA>(d3_dpp<0x104, 0x5>(v, v), v);
    else if (J == 8) r = d3_dpp<0x118, 0xC>(d3_dpp<0x108, 0x3>(v, v), v);
    else if (J == 16) { const auto p = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false); r = (int)((lane & 16) ? p[0] : p[1]); }
    else { const auto p = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false); r = (int)((lane & 32) ? p[0] : p[1]); }
    return __int_as_float(r);
}
// the sum of one double per lane, in every lane: a butterfly (partners 1, 2, 4, ... 32 apart add up -- the same pairs in either order, so all
// lanes end with the same bits)
template <int J>
__device__ static inline double lane_xor_f64(double v, int lane)
{
    const float lo = lane_xor<J>(__int_as_float(__double2loint(v)), lane), hi = lane_xor<J>(__int_as_float(__double2hiint(v)), lane);
    return __hiloint2double(__float_as_int(hi), __float_as_int(lo));
}
__device__ static inline double wave_sum_f64(double v, int lane)
{
    v += lane_xor_f64<1>(v, lane); v += lane_xor_f64<2>(v, lane); v += lane_xor_f64<4>(v, lane);
    v += lane_xor_f64<8>(v, lane); v += lane_xor_f64<16>(v, lane); v += lane_xor_f64<32>(v, lane);
    return v;
}
// inclusive scan of one int per lane (row shifts, then the row broadcasts of gfx9)
__device__ static inline int wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);              // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);              // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);              // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);              // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);              // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);              // row_bcast:31 into rows 2 and 3
    return v;
}
// lanes of a wave that hold the same cell code (< 0: none): the lowest such lane, how many there are, and this lane's rank
// among them -- so that ONE lane per (wave, cell) talks to memory, and all of a wave's leaders do so in one instruction
// (arrival order runs along the rings: neighbours hit the same post, i.e. the same cell; a returning atomic per distinct
// cell in turn cost a wave ten dependent round trips)
__device__ static inline void wave_groups(int code, int lane, int &leader, int &count, int &rank)
{
    leader = lane; count = 0; rank = 0;
    unsigned long long todo = __ballot(code >= 0);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        const int c0 = __builtin_amdgcn_readlane(code, src);
        const unsigned long long grp = __ballot(code == c0);
        if (code == c0) { leader = src; count = __popcll(grp); rank = __popcll(grp & ((1ull << lane) - 1)); }
        todo &= ~grp;
    }
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
// Round 4: the survivors' grid-cell histogram is taken beside the count, the cell scan beside the write.
__global__ __launch_bounds__(1024) void k3_filter_count(Det3dBufs B, int N, double intensity_min)
{
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * 1024 + tid;
    const float4 cur = (i < N) ? ((const float4 *)B.xyzi)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool keep = i < N && (double)cur.w > intensity_min;      // :33
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) wsum[wave] = __popcll(bal);
    if (bal) {
        const int code = keep ? cell_code(cur.x, cur.y, B.ctl->gx0, B.ctl->gy0, B.ctl->ginv) : -1;
        int leader, count, rank;
        wave_groups(code, lane, leader, count, rank);
        if (keep && lane == leader) atomicAdd(&B.hist[code], count);
    }
    __syncthreads();
    if (tid == 0) {
        int c = 0;
        for (int w = 0; w < 16; ++w) c += wsum[w];
        B.cnt[blockIdx.x] = c;                                                   // B.cnt is rebuilt by k3_scatter for its own use
    }
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
    const int i = blockIdx.x * 1024 + tid;
    const float4 cur = (i < N) ? ((const float4 *)B.xyzi)[i] : make_float4(0.f, 0.f, 0.f, 0.f);   // (in flight under the counts)
    if (tid < 64) {                                                              // survivors in the tiles before this one
        int c = 0;
        for (int w = tid; w < (int)blockIdx.x; w += 64) c += B.cnt[w];
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
        if (tid == 0) base = c;
    }
    __syncthreads();
    const bool keep = i < N && (double)cur.w > intensity_min;
    const int pos = tile_compact_pos(keep, wsum, &base);
    if (keep) { B.p1[pos] = cur.x; B.p1[B.cap + pos] = cur.y; B.p1[2 * B.cap + pos] = cur.z; }
    if ((int)blockIdx.x == ftiles - 1 && tid == 0) { B.ctl->M = base; B.ctl->M2 = 0; B.ctl->K = 0; B.ctl->err = 0; B.ctl->nroots = 0; B.ctl->retry = 0; B.ctl->m_true = base; }   // base now includes this tile
}

// Reductions over the lanes with the partner taken straight through DPP: ONE instruction a level (row shifts 1, 2, 4, 8, then the row
// broadcasts 15 and 31), written out because the compiler turns the builtins into copy + wait + v_mov_b32_dpp + operation, and fminf / fmaxf
// into that plus a canonicalising v_max.  A lane whose source lies outside its row (or whose row is masked) keeps its value; the result is
// lane 63's (all lanes), lane 31's (the lower half), lane 15's of every row (that row).
#define D3_DPP_LEVEL(OP, V, CTRL) asm volatile("s_nop 1\n\t" OP " %0, %0, %0 " CTRL : "+v"(V))
#define D3_DPP_REDUCE32(OP, V)                                                                                     \
    D3_DPP_LEVEL(OP, V, "row_shr:1 row_mask:0xf bank_mask:0xf"); D3_DPP_LEVEL(OP, V, "row_shr:2 row_mask:0xf bank_mask:0xf"); \
    D3_DPP_LEVEL(OP, V, "row_shr:4 row_mask:0xf bank_mask:0xf"); D3_DPP_LEVEL(OP, V, "row_shr:8 row_mask:0xf bank_mask:0xf"); \
    D3_DPP_LEVEL(OP, V, "row_bcast:15 row_mask:0xa bank_mask:0xf")
// the smallest of one non-negative int per lane (as unsigned: also of non-negative floats' bit patterns), in every lane
__device__ static inline int wave_min_u32(int m)
{
    D3_DPP_REDUCE32("v_min_u32_dpp", m);
    D3_DPP_LEVEL("v_min_u32_dpp", m, "row_bcast:31 row_mask:0xc bank_mask:0xf");
    asm volatile("s_nop 1" ::: "memory");
    return __builtin_amdgcn_readlane(m, 63);
}

// bounding boxes of BOX_PTS consecutive sorted points: lane = point (x, y, z in registers), 32-lane halves reduce by
// shuffles.  NaN coordinates (the outliers k3_cc_min masks: x) are skipped (v_min_f32 / v_max_f32 return the other operand); a tile without any number gets an
// empty box (+inf, -inf).
__device__ static inline void tile_boxes(float *box, int s, int M, float x, float y, float z)
{
    const float qn = __int_as_float(0x7fc00000);
    const bool ok = s < M && x == x;                                              // x carries the mask
    // (v_min_f32 / v_max_f32 skip a QUIET NaN; a signalling one -- whatever bits the driver sent -- would come back as the result: every NaN is
    // replaced by the quiet one first)
    const float cx = ok ? x : qn, cy = (ok && y == y) ? y : qn, cz = (ok && z == z) ? z : qn;
    float m[6] = {cx, cy, cz, cx, cy, cz};
#pragma unroll
    for (int k = 0; k < 3; ++k) { D3_DPP_REDUCE32("v_min_f32_dpp", m[k]); D3_DPP_REDUCE32("v_max_f32_dpp", m[3 + k]); }
    asm volatile("s_nop 1" ::: "memory");
    if ((threadIdx.x & 31) == 31 && (s & ~31) < M) {                              // (the half's last lane holds its minimum / maximum)
        float4 lo, hi;
        lo.x = (m[0] == m[0]) ? m[0] : INFINITY; lo.y = (m[1] == m[1]) ? m[1] : INFINITY; lo.z = (m[2] == m[2]) ? m[2] : INFINITY;
        lo.w = (m[3] == m[3]) ? m[3] : -INFINITY; hi.x = (m[4] == m[4]) ? m[4] : -INFINITY; hi.y = (m[5] == m[5]) ? m[5] : -INFINITY;
        hi.z = hi.w = 0.f;
        *(float4 *)(box + 8 * (s >> 5)) = lo; *(float4 *)(box + 8 * (s >> 5) + 4) = hi;
    }
}

// ---- the counting sort's scatter (thread = node): sorted coordinates + the permutation; also clears what the next
// stages and the next call expect cleared
__global__ __launch_bounds__(1024) void k3_scatter(Det3dBufs B, int N)
{
    // (every first load is asked for at once -- M, the grid, the node's coordinates whether or not it exists -- and waited for once:
    // right behind a kernel boundary each dependent load is a trip to memory, ~1 us)
    const int M = B.ctl->M;
    const float gx0 = B.ctl->gx0, gy0 = B.ctl->gy0, ginv = B.ctl->ginv;
    const int gid = blockIdx.x * 1024 + threadIdx.x, lane = threadIdx.x & 63;
    const bool v0 = gid < N;
    const float x = v0 ? B.p1[gid] : 0.f, y = v0 ? B.p1[B.cap + gid] : 0.f, z = v0 ? B.p1[2 * B.cap + gid] : 0.f;
    for (int k = gid; k < GRID_CELLS; k += gridDim.x * 1024) B.hist[k] = 0;
    if ((int)blockIdx.x * 1024 >= M) return;
    const bool v = gid < M;
    const int code = v ? cell_code(x, y, gx0, gy0, ginv) : -1;
    int leader, count, rank;
    wave_groups(code, lane, leader, count, rank);
    int base = 0;
    if (v && lane == leader) base = atomicAdd(&B.cursor[code], count);        // one cursor bump per (wave, cell), all of them in flight together
    base = __shfl(base, leader, 64);
    if (!v) return;
    const int pos = base + rank;
    B.s1[pos] = x; B.s1[B.cap + pos] = y; B.s1[2 * B.cap + pos] = z;
    B.perm[pos] = gid;
    B.cnt[gid] = 0; B.first[gid] = 0x7fffffff; B.last[gid] = 0;
}
__global__ __launch_bounds__(256) void k3_boxes(Det3dBufs B, int N)
{
    const int M = B.ctl->M;
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int ss = s < N ? s : 0;
    const float x = B.s1[ss], y = B.s1[B.cap + ss], z = B.s1[2 * B.cap + ss];    // (beside M, not behind it; past M: masked)
    if ((int)blockIdx.x * 256 >= M) return;
    tile_boxes(B.box, s, M, x, y, z);
}

// ---- the short front end (round 6): gate + sort + boxes in ONE launch (k3f_front; or two: k3f_gate + k3f_sort, -DD3_TWO_LAUNCH_FRONT) for clouds
// whose survivors fit one workgroup's LDS ------------
// The four launches above are a chain of all-to-all hand-overs of almost no data (3 k survivors = 40 KB): 20 us.  Here every 1024-point
// tile compacts its own survivors in place (k3f_gate: no count to wait for, no histogram), and ONE workgroup does the rest in LDS
// (f_sort_body): the tiles' prefix, the survivors in registers (up to five per thread, asked for together), the histogram of a 64 x 64
// grid's cells by LDS atomics whose return value is the rank inside the cell, the scan of the cells, the sorted copy (over the histogram's
// storage), boxes, and coalesced stores.  Node numbers are the same as the long chain's (tile prefix + place in the tile = arrival index among the survivors),
// the order inside a cell is as arbitrary as there: nothing of the result can tell the two front ends apart.
constexpr int MFAST = 5120, MFAST_PT = MFAST / 1024;   // (at 6.5 k survivors one workgroup's sort and the LDS forest are no faster than the long chain; at 3.3 k they save 7 us)
constexpr int FGRID_G = 64, FGRID_CELLS = FGRID_G * FGRID_G;     // k3f_sort's grid: at most two points per cell on average, a quarter of the cells to clear and scan
#ifdef D3_TWO_LAUNCH_FRONT
__global__ __launch_bounds__(1024) void k3f_gate(Det3dBufs B, int N, double intensity_min)
{
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * 1024 + tid;
    const float4 cur = (i < N) ? ((const float4 *)B.xyzi)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool keep = i < N && (double)cur.w > intensity_min;      // :33
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int c = wsum[w]; if (w < wave) off += c; tot += c; }
    if (keep) {
        const int pos = blockIdx.x * 1024 + off + __popcll(bal & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
        B.p1[pos] = cur.x; B.p1[B.cap1 + pos] = cur.y; B.p1[2 * B.cap1 + pos] = cur.z;
    }
    if (tid == 0) B.cnt[blockIdx.x] = tot;
}
#endif
#define D3_KERNEL 4
// what another workgroup of the SAME launch has written (k3f_front): past this XCD's L2
template <bool DEV, typename T>
__device__ static inline T f_ld(const T *p) { return DEV ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; }
template <bool DEV>
__device__ static inline void f_sort_body(const Det3dBufs &B, int N, int ftiles)
{
    __shared__ int s_pre[1024 + 1];
    __shared__ int s_w[16];
    __shared__ __attribute__((aligned(16))) int s_mem[4 * MFAST];                  // the cell histogram (4 k ints), then the sorted copy: x | y | z | node
    static_assert(FGRID_CELLS <= 4 * MFAST && FGRID_CELLS == 4 * 1024, "k3f_sort's LDS plan");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    D3_MARK(0);
    const int c = (tid < ftiles) ? f_ld<DEV>(&B.cnt[tid]) : 0;
    const float gx0 = B.ctl->gx0, gy0 = B.ctl->gy0, ginv = B.ctl->ginv;
#pragma unroll
    for (int k = 0; k < FGRID_CELLS / 1024; ++k) s_mem[tid + 1024 * k] = 0;
    // exclusive scan of the tiles' counts
    int incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int wbase = 0, M = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int v = s_w[w]; if (w < wave) wbase += v; M += v; }
    s_pre[tid] = wbase + incl - c;
    if (tid == 0) s_pre[1024] = M;
    __syncthreads();
    if (M > MFAST) {                                                              // (uniform) not for this front end
        if (tid == 0) { B.ctl->M = 0; B.ctl->M2 = 0; B.ctl->K = 0; B.ctl->err = 0; B.ctl->nroots = 0; B.ctl->retry = 1; B.ctl->m_true = M; }
        return;
    }
    D3_MARK(1);
    // the survivors, MFAST_PT per thread: node g sits in tile t = the last one whose prefix is <= g, at place g - prefix
    float x[MFAST_PT], y[MFAST_PT], z[MFAST_PT];
    int code[MFAST_PT], rank[MFAST_PT];
    {
        int steps = 0;
        while ((1 << steps) < ftiles) ++steps;
        int t[MFAST_PT];
#pragma unroll
        for (int r = 0; r < MFAST_PT; ++r) t[r] = 0;
        for (int b = steps - 1; b >= 0; --b) {                                    // (all of a thread's searches side by side: one LDS round trip per bit)
#pragma unroll
            for (int r = 0; r < MFAST_PT; ++r) {
                if (1024 * r >= M) break;                                         // (whole rounds nobody has an item in: skipped by every wave)
                const int u = t[r] | (1 << b);
                if (u < ftiles && s_pre[u] <= tid + 1024 * r) t[r] = u;
            }
        }
#pragma unroll
        for (int r = 0; r < MFAST_PT; ++r) {
            const int g = tid + 1024 * r;
            x[r] = y[r] = z[r] = 0.f;
            if (g < M) {                                                          // (a load nobody needs still costs its cycles of the CU's one memory pipeline)
                const int idx = t[r] * 1024 + (g - s_pre[t[r]]);
                x[r] = f_ld<DEV>(&B.p1[idx]); y[r] = f_ld<DEV>(&B.p1[B.cap1 + idx]); z[r] = f_ld<DEV>(&B.p1[2 * B.cap1 + idx]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < MFAST_PT; ++r) {
        const int g = tid + 1024 * r;
        code[r] = rank[r] = 0;
        if (g < M) {
            code[r] = cell_code<FGRID_G>(x[r], y[r], gx0, gy0, ginv);
            rank[r] = atomicAdd(&s_mem[code[r]], 1);
        }
        if (r == 0) D3_MARK(2);
    }
    __syncthreads();
    D3_MARK(3);
    // exclusive scan of the 4 k cells: wave w takes the cells [256 w, + 256) in four rounds of 64 consecutive ones (conflict-free
    // LDS accesses, the scan over the lanes on DPP); the chunks' bases stay apart (s_w) and are added when a cell is looked up.  (Sixteen
    // consecutive cells per THREAD, the first version, was a 16-way bank conflict on every access: 4 us.)
    {
        int carry = 0;
#pragma unroll
        for (int k = 0; k < FGRID_CELLS / 1024; ++k) {
            const int a = (FGRID_CELLS / 16) * wave + 64 * k + lane;
            const int v = s_mem[a];
            const int inc = wave_incl_scan(v);
            s_mem[a] = carry + inc - v;
            carry += __builtin_amdgcn_readlane(inc, 63);
        }
        if (lane == 0) s_w[wave] = carry;                                         // (two barriers since anybody read the tiles' sums)
    }
    __syncthreads();
    int cbase = 0;                                                                // lane w: the cells in front of chunk w
    {
        const int v = (lane < 16) ? s_w[lane] : 0;
        cbase = wave_incl_scan(v) - v;
    }
    D3_MARK(4);
    int pos[MFAST_PT];
#pragma unroll
    for (int r = 0; r < MFAST_PT; ++r) {
        pos[r] = 0;
        if (1024 * r < M) pos[r] = s_mem[code[r]] + __shfl(cbase, code[r] / (FGRID_CELLS / 16), 64) + rank[r];
    }
    __syncthreads();
    float *sx = reinterpret_cast<float *>(s_mem), *sy = sx + MFAST, *sz = sx + 2 * MFAST;
    int *sp = s_mem + 3 * MFAST;
#pragma unroll
    for (int r = 0; r < MFAST_PT; ++r) {
        const int g = tid + 1024 * r;
        if (g < M) { sx[pos[r]] = x[r]; sy[pos[r]] = y[r]; sz[pos[r]] = z[r]; sp[pos[r]] = g; }
    }
    __syncthreads();
    D3_MARK(5);
    for (int s0 = 0; s0 < M; s0 += 1024) {
        const int sidx = s0 + tid;
        const bool v = sidx < M;
        const float px = v ? sx[sidx] : 0.f, py = v ? sy[sidx] : 0.f, pz = v ? sz[sidx] : 0.f;
        if (v) {
            B.s1[sidx] = px; B.s1[B.cap + sidx] = py; B.s1[2 * B.cap + sidx] = pz;
            B.perm[sidx] = sp[sidx];
        }
        tile_boxes(B.box, sidx, M, px, py, pz);
    }
    if (tid == 0) { B.ctl->M = M; B.ctl->M2 = 0; B.ctl->K = 0; B.ctl->err = 0; B.ctl->nroots = 0; B.ctl->retry = 0; B.ctl->m_true = M; }
    D3_MARK(6);
}
#ifdef D3_TWO_LAUNCH_FRONT
__global__ __launch_bounds__(1024) void k3f_sort(Det3dBufs B, int N, int ftiles) { f_sort_body<false>(B, N, ftiles); }
#endif
// ... and both in ONE launch: the tile workgroup that counts itself in last goes on as the sorting workgroup.  No workgroup waits for
// another: a tile's survivors and count are written through to memory (device-scope stores), waited for, and only then does the tile's
// first thread bump the arrival counter; whoever sees ftiles - 1 there knows everything has landed and reads it past its own L2
// (device-scope loads).  One kernel boundary and one kernel's start-up less than k3f_gate + k3f_sort.
__global__ __launch_bounds__(1024) void k3f_front(Det3dBufs B, int N, double intensity_min, int ftiles)
{
    __shared__ int wsum[16];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * 1024 + tid;
    const float4 cur = (i < N) ? ((const float4 *)B.xyzi)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool keep = i < N && (double)cur.w > intensity_min;      // :33
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int c = wsum[w]; if (w < wave) off += c; tot += c; }
    if (keep) {
        const int pos = blockIdx.x * 1024 + off + __popcll(bal & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
        __hip_atomic_store(&B.p1[pos], cur.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&B.p1[B.cap1 + pos], cur.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&B.p1[2 * B.cap1 + pos], cur.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) __hip_atomic_store(&B.cnt[blockIdx.x], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             // every wave: its stores have been acknowledged
    __syncthreads();
    if (tid == 0) {
        const int before = __hip_atomic_fetch_add(&B.ctl->arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = before == ftiles - 1;
        if (s_last) __hip_atomic_store(&B.ctl->arrived, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (for the next cloud)
    }
    __syncthreads();
    if (!s_last) return;
    f_sort_body<true>(B, N, ftiles);
}
#undef D3_KERNEL

// ---- the three neighbour sweeps: ONE WAVE PER QUERY, lane = candidate ----------------------------------------
// (Round 4.  Earlier forms, all exact, all parity-green: lane = query with the candidates through the scalar cache and a
// sorted list per lane -- 142 us all-pairs, 104 us with the boxes below; the same with LDS-staged tiles, per-lane tile
// selection and bitonic merges on register arrays -- 42 us: the work of a workgroup of 64 queries is what its neediest
// lane needs, and 56 workgroups leave 200 CUs idle.)  A wave takes one query of the sorted copy.  Its 64 lanes hold 64
// candidates: the aligned 64 points around the query first, then every tile of 32 whose BOX lies within the query's
// current bound (all boxes are tested 64 at a time, one per lane), two tiles per step, four steps' loads in flight.  No
// LDS, no barrier, every load coalesced, a few thousand independent waves: the whole chip works, and each wave only on
// what its own query needs.
constexpr int KNN = MEAN_K + 1;

// one compare-exchange stage of a bitonic network over the lanes: blocks of K lanes alternate direction (K = 64: one
// block), partners are J apart.  min or max as ONE v_med3_f32 against -inf / +inf (no NaN ever enters).
template <int K, int J, bool DESC>
__device__ static inline float bitonic_stage(float v, int lane)
{
    const float o = lane_xor<J>(v, lane);
    const bool asc = (K == 64) ? !DESC : (((lane & K) == 0) != DESC);
    return __builtin_amdgcn_fmed3f(v, o, (asc == ((lane & J) == 0)) ? -INFINITY : INFINITY);
}
// sort one value per lane: ascending by lane, or descending
template <bool DESC>
__device__ static inline float wave_sort64(float v, int lane)
{
#define D3_STG(K, J) v = bitonic_stage<K, J, DESC>(v, lane);
    D3_STG(2, 1)
    D3_STG(4, 2) D3_STG(4, 1)
    D3_STG(8, 4) D3_STG(8, 2) D3_STG(8, 1)
    D3_STG(16, 8) D3_STG(16, 4) D3_STG(16, 2) D3_STG(16, 1)
    D3_STG(32, 16) D3_STG(32, 8) D3_STG(32, 4) D3_STG(32, 2) D3_STG(32, 1)
    D3_STG(64, 32) D3_STG(64, 16) D3_STG(64, 8) D3_STG(64, 4) D3_STG(64, 2) D3_STG(64, 1)
#undef D3_STG
    return v;
}
// S ascending, D DESCENDING by lane -> the 64 smallest of both, ascending by lane
__device__ static inline float wave_merge64(float S, float D, int lane)
{
    float c = fminf(S, D);                                        // bitonic: an ascending run against a descending one
    c = bitonic_stage<64, 32, false>(c, lane); c = bitonic_stage<64, 16, false>(c, lane); c = bitonic_stage<64, 8, false>(c, lane);
    c = bitonic_stage<64, 4, false>(c, lane); c = bitonic_stage<64, 2, false>(c, lane); c = bitonic_stage<64, 1, false>(c, lane);
    return c;
}
// the lane with the smallest v among the lanes of `set` (nearest tile first: the bound tightens at once and most of the
// other tiles are never opened)
__device__ static inline int nearest_of(unsigned long long set, float v, int lane)
{
    // (v >= 0 or +inf: the bit patterns order like the values)
    const bool in = (set >> lane) & 1ull;
    const int mn = wave_min_u32(in ? __float_as_int(v) : 0x7f800000);
    return __ffsll((long long)__ballot(in && __float_as_int(v) == mn)) - 1;
}
constexpr int QW = 4;               // queries (waves) per workgroup
constexpr int Q_GRID = 2048;        // workgroups: the queries are dealt round-robin
#ifndef D3_KNN_AHEAD
#define D3_KNN_AHEAD 3
#endif
#ifndef D3_CCMIN_AHEAD
#define D3_CCMIN_AHEAD 4
#endif
constexpr int KNN_AHEAD = D3_KNN_AHEAD;     // steps (pairs of tiles) whose loads are issued together
constexpr int CC_AHEAD = 4;                 // the same in k3_cc_link's sweep
constexpr int CCMIN_AHEAD = D3_CCMIN_AHEAD; // ... and in k3_cc_min's
#ifndef D3_KNN_FEW
#define D3_KNN_FEW 6
#endif
constexpr int KNN_FEW = D3_KNN_FEW;      // a step with at most this many admissible candidates inserts them one by one

// What a sweep can ask for before it knows anything but its query's number: the boxes of the first 128 tiles (two per lane).  All of a
// query's first loads -- M, the query, its aligned 64 neighbours, these boxes -- are issued together and waited for once (each used to
// wait for the one before: four round trips to a memory that is a microsecond away right after a kernel boundary).
struct BoxPre { float4 lo0, hi0, lo1, hi1; };
__device__ static inline void box_prefetch(const float *box, int lane, int ntiles_ub, BoxPre &P)
{
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    P.lo0 = P.hi0 = P.lo1 = P.hi1 = z;
    if (lane < ntiles_ub) { P.lo0 = *(const float4 *)(box + 8 * lane); P.hi0 = *(const float4 *)(box + 8 * lane + 4); }
    if (lane + 64 < ntiles_ub) { P.lo1 = *(const float4 *)(box + 8 * (lane + 64)); P.hi1 = *(const float4 *)(box + 8 * (lane + 64) + 4); }
}
// squared distance from the query to the box of tile r0 + lane (inf past the last tile)
__device__ static inline float round_box_d2(const float *box, const BoxPre &P, int r0, int lane, int ntiles, float qx, float qy, float qz)
{
    const int t = r0 + lane;
    if (t >= ntiles) return INFINITY;
    float4 lo, hi;
    if (r0 == 0) { lo = P.lo0; hi = P.hi0; }
    else if (r0 == 64) { lo = P.lo1; hi = P.hi1; }
    else { lo = *(const float4 *)(box + 8 * t); hi = *(const float4 *)(box + 8 * t + 4); }
    return box_d2(qx, qy, qz, lo.x, lo.y, lo.z, lo.w, hi.x, hi.y);
}

// ---- SOR part 1: mean distance to the MeanK nearest neighbours (:43-47).  The 64 smallest squared distances seen so
// far live one per lane, ascending.  A step with many admissible candidates sorts its 64 across the lanes and merges
// (bitonic networks on DPP / permlane swaps, one v_med3 per stage); one with few inserts them one at a time (ballot,
// popcount, wave_shr:1).  The bound is lane 30's value: a tile is opened only while its box is nearer than that.  The
// multiset of the 31 smallest values is exact, so the ascending-order FP64 sum is bit-identical to the reference's.
// m_hint: the previous cloud's M (clouds of one sensor look alike).  A wave whose query number is below it asks for everything at once;
// one above it (most of the grid: the grid is sized for N, the gate leaves an eighth) first waits for M -- 5000 idle waves asking for
// eleven loads each cost the working ones 6 us.  A wrong hint costs time, never a result.
#define D3_KERNEL 2
__global__ __launch_bounds__(64 * QW) void k3_knn(Det3dBufs B, int N, int ntiles_ub, int m_hint)
{
    D3_MARK(0);
    const int lane = threadIdx.x & 63;
    const int M = B.ctl->M;                                                       // (the first load out: waited for behind the query's own)
    const float *__restrict__ X = B.s1, *__restrict__ Y = B.s1 + B.cap, *__restrict__ Z = B.s1 + 2 * B.cap;
    for (int q = __builtin_amdgcn_readfirstlane(blockIdx.x * QW + (threadIdx.x >> 6)); q < N; q += gridDim.x * QW) {
        if (q >= m_hint && q >= M) break;
        const float qx = X[q], qy = Y[q], qz = Z[q];
        const int a0 = q & ~63;                                  // the aligned 64 points around the query = tiles a0/32, a0/32 + 1
        const int ja = a0 + lane;
        const float ax = X[ja], ay = Y[ja], az = Z[ja];          // (past M: padding, masked)
        const int node = B.perm[q];
        BoxPre P;
        box_prefetch(B.box, lane, ntiles_ub, P);
        if (q >= M) break;
        D3_MARK(1);
        int dbg_steps = 0, dbg_merges = 0;
        (void)dbg_steps; (void)dbg_merges;
        const int ntiles = (M + BOX_PTS - 1) / BOX_PTS;
        float S;
        {
            const float d2 = d2f(qx, qy, qz, ax, ay, az);
            S = wave_sort64<false>((ja < M && d2 == d2) ? d2 : INFINITY, lane);
        }
        float bound = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(S), KNN - 1));
        D3_MARK(2);
        for (int r0 = 0; r0 < ntiles; r0 += 64) {
            const int t = r0 + lane;
            const float db = ((t >> 1) == (a0 >> 6)) ? INFINITY : round_box_d2(B.box, P, r0, lane, ntiles, qx, qy, qz) * BOX_MARGIN;
            unsigned long long todo = __ballot(db < bound);
            while (todo) {
                float cx[KNN_AHEAD], cy[KNN_AHEAD], cz[KNN_AHEAD];
                bool ok[KNN_AHEAD];
                int nstep = 0;
#pragma unroll
                for (int u = 0; u < KNN_AHEAD; ++u) {
                    int ta = -1, tb = -1;
                    if (todo) { ta = nearest_of(todo, db, lane); todo &= ~(1ull << ta); ++nstep; }
                    if (todo) { tb = nearest_of(todo, db, lane); todo &= ~(1ull << tb); }
                    const int tile = (lane < 32) ? ta : tb;
                    const int j = BOX_PTS * (r0 + tile) + (lane & 31);
                    ok[u] = tile >= 0 && j < M;
                    const int jj = ok[u] ? j : 0;
                    cx[u] = X[jj]; cy[u] = Y[jj]; cz[u] = Z[jj];
                }
#pragma unroll
                for (int u = 0; u < KNN_AHEAD; ++u) {
                    if (u >= nstep) break;
                    ++dbg_steps;
                    const float d2 = d2f(qx, qy, qz, cx[u], cy[u], cz[u]);
                    const float d = (ok[u] && d2 == d2) ? d2 : INFINITY;
                    unsigned long long adm = __ballot(d < bound);
                    if (adm == 0ull) continue;
                    if (__popcll(adm) <= KNN_FEW) {
                        while (adm) {
                            const int l = __ffsll((long long)adm) - 1;
                            adm &= adm - 1;
                            const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), l));
                            if (!(x < bound)) continue;
                            const int pos = __popcll(__ballot(S <= x));           // S is ascending: a prefix of the lanes keeps its place
                            const float sh = __int_as_float(d3_dpp<0x138, 0xf>(__float_as_int(S), __float_as_int(S)));   // wave_shr:1
                            S = (lane < pos) ? S : ((lane == pos) ? x : sh);
                            bound = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(S), KNN - 1));
                        }
                    } else {
                        ++dbg_merges;
                        S = wave_merge64(S, wave_sort64<true>(d, lane), lane);
                        bound = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(S), KNN - 1));
                    }
                }
                todo &= __ballot(db < bound);
            }
        }
        D3_MARK(3);
        D3_NOTE(6, dbg_steps); D3_NOTE(7, dbg_merges);
        const float sq = sqrtf(S);
        double dist_sum = 0;
#pragma unroll
        for (int k = 1; k < KNN; ++k) dist_sum += (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(sq), k));   // k = 0 is the query itself
        if (lane == 0) {
            const float md = (M >= KNN) ? (float)(dist_sum / MEAN_K) : 0.f;      // fewer than MeanK+1 points: the search "failed"
            B.dist[node] = md; B.dist_s[q] = md;                    // (by node: the statistics' order; by sorted position: the sweeps')
        }
        D3_MARK(4);
    }
}
#undef D3_KERNEL

// ---- connected components of the radius graph: lock-free union-find ------------------------------
// parent = B.label.  Only roots are ever hooked (CAS root -> a SMALLER root), so the final root of a
// component is its smallest index whatever the interleaving: deterministic labels from one all-pairs pass
// (the first version needed up to 64 propagation launches).  The XCDs' L2 caches are not coherent with each other
// for plain loads, and a device-scope atomic load is a ~1.5 us round trip, so the finds read parent[] through the
// caches: every value parent[x] has ever held is an ancestor of x for good (hooks attach roots under smaller indices,
// halving only shortcuts upwards), hence a stale read can only return an ancestor that is no longer the root -- never a
// wrong one.  What must be exact is the hook itself: the CAS on the larger root goes to memory, and when that node has stopped
// being a root it fails AND returns the node's parent as memory has it -- the walk goes on from there (round 5; round 4 repeated
// both finds with device-scope loads: the 10 us tail of the slowest waves).
// parent[] as the caches have it: a plain load the compiler may neither tear nor invent (wavefront-scope relaxed: no cache-control bits).
// Round 4 read it `volatile` -- which is a SYSTEM-scope load on this target: every one of them past the L1, and the few hundred waves
// of a big cluster, all asking for the same few parents, queued up on the same lines for 10 us.
__device__ static inline int uf_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ static int uf_find(int *parent, int x)
{
    int p = uf_ld(&parent[x]);
    while (p != x) {
        const int gp = uf_ld(&parent[p]);
        if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // path halving
        x = p; p = gp;
    }
    return x;
}
// joins the trees of a and b; returns the root as far as this thread knows it
__device__ static int uf_union(int *parent, int a, int b)
{
    a = uf_find(parent, a); b = uf_find(parent, b);
    while (a != b) {
        const int hi = max(a, b), lo = min(a, b);
        const int old = atomicCAS(&parent[hi], hi, lo);
        if (old == hi) return lo;
        const int up = uf_find(parent, old);                     // hi has a parent (< hi): on from there
        if (hi == a) a = up; else b = up;
    }
    return a;
}
// Two passes.  Doing every union inside one sweep serialises on chains of finds and compare-and-swaps.  Instead:
//   k3_cc_min   registers only: parent[i] = smallest index among i and its neighbours.  That alone puts nearly
//               every point of a compact cluster in one tree (the chains run towards the cluster's first point).  Round 5: also
//               SOR part 2 -- a query farther than the threshold gets parent -1, a candidate farther than it is no neighbour
//               (round 4: a kernel of its own, k3_sor, that masked the outliers' x and rebuilt the boxes without them);
//   k3_cc_link  every neighbour found earlier in the sorted copy is looked up with the top of its chain as it stands (equal
//               tops mean "same tree already", and a stale top is still an ancestor, so the filter never drops a needed
//               union), and only adjacent pairs whose tops differ go into the union code -- once per DISTINCT top of a query's
//               neighbours, by one lane (round 4: every lane for itself, dozens of compare-and-swaps on the same root).
// Both as one wave per query over the tiles whose box lies within 0.2 m of it, like k3_knn.
#define D3_KERNEL 3
__global__ __launch_bounds__(64 * QW) void k3_cc_min(Det3dBufs B, int N, int ntiles_ub, int m_hint)
{
    __shared__ double red[2][QW];
    D3_MARK(0);
    const int M = B.ctl->M;
    const int tid = threadIdx.x, lane = tid & 63;
    const float *__restrict__ X = B.s1, *__restrict__ Y = B.s1 + B.cap, *__restrict__ Z = B.s1 + 2 * B.cap, *__restrict__ DS = B.dist_s;
    // the first query's loads go out before the statistics
    int q = __builtin_amdgcn_readfirstlane(blockIdx.x * QW + (threadIdx.x >> 6));
    if ((int)blockIdx.x * QW >= m_hint && (int)blockIdx.x * QW >= M) return;      // (the whole workgroup: no query; k3_knn on m_hint)
    const int q0 = q < N ? q : 0;
    float qx = X[q0], qy = Y[q0], qz = Z[q0], qd = DS[q0];
    int own = B.perm[q0];
    // every 64th query also looks after the boxes of the two tiles it starts: see below
    const bool boxer = (q & 63) == 0 && q < N;
    float tx = 0.f, ty = 0.f, tz = 0.f, td = 0.f;
    if (boxer) { tx = X[q + lane]; ty = Y[q + lane]; tz = Z[q + lane]; td = DS[q + lane]; }
    BoxPre P;
    box_prefetch(B.box, lane, ntiles_ub, P);
    // ... and so do the first 16 * 64 * QW of the M distances the statistics are taken over (sixteen per thread, by node; the arrays end 1024
    // floats behind B.cap): with everything else a workgroup asks for before it knows M.  (Round 5 waited for M, then for one distance after
    // the other: 5 us of this kernel's 12.)
    constexpr int SPT = 16, SPW = SPT * 64 * QW;                                  // distances per thread and round / per workgroup and round
    float4 e[SPT / 4];
    {
        const bool can = SPT * tid + SPT - 1 < B.cap + 1024;
#pragma unroll
        for (int k = 0; k < SPT / 4; ++k) e[k] = can ? ((const float4 *)B.dist)[(SPT / 4) * tid + k] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if ((int)blockIdx.x * QW >= M) return;                                        // (the whole workgroup: no query)
    D3_MARK(1);
    // ---- SOR part 2 (:43-47 setStddevMulThresh): mean and (n-1)-variance of the M distances in FP64.  EVERY workgroup takes them for
    // itself (M floats out of L2: cheaper than a launch in between, and than any hand-over inside one), in an order that depends on M alone:
    // thread t adds up the distances of the nodes [16 (t + 256 r), + 16), r = 0, 1, ..., one after the other; the threads' sums meet in a
    // butterfly over the lanes and the four waves' in turn -- the threshold's bits do not depend on the grid or on which workgroup asks.
    // Which points are outliers is never stored: the sweeps compare dist_s with the threshold as they go.
    double thr;
    {
        double sum = 0, sq = 0;
        for (int base = SPT * tid; base < M; base += SPW) {
            if (base != SPT * tid) {
#pragma unroll
                for (int k = 0; k < SPT / 4; ++k) e[k] = ((const float4 *)B.dist)[base / 4 + k];
            }
#pragma unroll
            for (int k = 0; k < SPT / 4; ++k) {
                const float v4[4] = {e[k].x, e[k].y, e[k].z, e[k].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double v = (base + 4 * k + j < M) ? (double)v4[j] : 0.0;
                    sum += v; sq += v * v;
                }
            }
        }
        sum = wave_sum_f64(sum, lane); sq = wave_sum_f64(sq, lane);
        if (lane == 0) { red[0][tid >> 6] = sum; red[1][tid >> 6] = sq; }
        __syncthreads();
        double r0 = red[0][0], r1 = red[1][0];
#pragma unroll
        for (int w = 1; w < QW; ++w) { r0 += red[0][w]; r1 += red[1][w]; }
        const double valid = (M >= MEAN_K + 1) ? (double)M : 0.0;
        const double mean = r0 / valid;
        const double variance = (r1 - r0 * r0 / valid) / (valid - 1);
        thr = mean + STD_MUL * sqrt(variance);                                   // (every thread: the same bits; NaN keeps everything)
    }
    D3_MARK(2);
    int dbg_tiles = 0;
    (void)dbg_tiles;
    for (bool first = true; q < N; q += gridDim.x * QW, first = false) {
        bool bx = boxer;
        if (!first) {
            if (q >= M) break;
            qx = X[q]; qy = Y[q]; qz = Z[q]; qd = DS[q];
            own = B.perm[q];
            bx = (q & 63) == 0;
            if (bx) { tx = X[q + lane]; ty = Y[q + lane]; tz = Z[q + lane]; td = DS[q + lane]; }
            box_prefetch(B.box, lane, ntiles_ub, P);
        }
        if (q >= M) break;
        // The boxes so far are those of ALL survivors of the gate; SOR's outliers -- a quarter of them, the sparse fringes -- blow them
        // up (twenty tiles within 0.2 m of a query instead of two).  The wave that starts a pair of tiles leaves their boxes without
        // the outliers in a second array: this kernel's sweeps run on the loose ones, k3_cc_link, a launch later, on the tight ones.
        // (Round 4: a kernel of its own.)
        if (bx) tile_boxes(B.box2, q + lane, M, ((double)td > thr) ? __int_as_float(0x7fc00000) : tx, ty, tz);
        if ((double)qd > thr) {                                  // :47 -- an outlier: out of the graph
            if (lane == 0) B.label[own] = -1;
            continue;
        }
        const int ntiles = (M + BOX_PTS - 1) / BOX_PTS;
        int mi = own;
        for (int r0 = 0; r0 < ntiles; r0 += 64) {
            unsigned long long todo = __ballot(round_box_d2(B.box, P, r0, lane, ntiles, qx, qy, qz) * BOX_MARGIN < TOL2);
            dbg_tiles += __popcll(todo);
            while (todo) {
                float cx[CCMIN_AHEAD], cy[CCMIN_AHEAD], cz[CCMIN_AHEAD], cd[CCMIN_AHEAD];
                int pj[CCMIN_AHEAD];
                bool ok[CCMIN_AHEAD];
#pragma unroll
                for (int u = 0; u < CCMIN_AHEAD; ++u) {                        // four steps' loads in flight
                    int ta = -1, tb = -1;
                    if (todo) { ta = __ffsll((long long)todo) - 1; todo &= todo - 1; }
                    if (todo) { tb = __ffsll((long long)todo) - 1; todo &= todo - 1; }
                    const int tile = (lane < 32) ? ta : tb;
                    const int j = BOX_PTS * (r0 + tile) + (lane & 31);
                    ok[u] = tile >= 0 && j < M;
                    const int jj = ok[u] ? j : 0;
                    cx[u] = X[jj]; cy[u] = Y[jj]; cz[u] = Z[jj]; cd[u] = DS[jj]; pj[u] = B.perm[jj];   // (the id with the coordinates, not behind the test)
                }
#pragma unroll
                for (int u = 0; u < CCMIN_AHEAD; ++u)
                    if (ok[u] && !((double)cd[u] > thr) && d2f(qx, qy, qz, cx[u], cy[u], cz[u]) < TOL2) mi = min(mi, pj[u]);
            }
        }
        mi = wave_min_u32(mi);                                   // (node numbers: >= 0)
        if (lane == 0) B.label[own] = mi;
        if (first) { D3_MARK(3); D3_NOTE(6, dbg_tiles); }
    }
}
#undef D3_KERNEL
#define D3_KERNEL 1
__global__ __launch_bounds__(64 * QW) void k3_cc_link(Det3dBufs B, int N, int ntiles_ub, int m_hint)
{
    const int M = B.ctl->M;
    const int lane = threadIdx.x & 63;
    const float *__restrict__ X = B.s1, *__restrict__ Y = B.s1 + B.cap, *__restrict__ Z = B.s1 + 2 * B.cap;
    int *parent = B.label;
    for (int q = __builtin_amdgcn_readfirstlane(blockIdx.x * QW + (threadIdx.x >> 6)); q < N; q += gridDim.x * QW) {
        if (q >= m_hint && q >= M) break;
        const float qx = X[q], qy = Y[q], qz = Z[q];
        const int own = B.perm[q];
        BoxPre P;
        box_prefetch(B.box2, lane, ntiles_ub, P);
        if (q >= M) break;
#ifdef RDET_DEBUG_MARKS
        const bool dbg = q == (int)blockIdx.x * QW;
        if (dbg) D3_MARK(0);
        unsigned long long dbg_tl = 0, dbg_tc = 0, dbg_tu = 0, dbg_t = 0;
        int dbg_batches = 0, dbg_hops = 0, dbg_rounds = 0, dbg_tiles = 0;
#endif
        int rs = uf_ld(&parent[own]);           // top of the query's chain as it stands
        if (rs < 0) continue;                                    // removed by SOR (k3_cc_min)
        for (int p = uf_ld(&parent[rs]); p != rs; p = uf_ld(&parent[rs])) rs = p;
        int ri = rs;                                             // (uniform) the query's root as far as this wave knows it
#ifdef RDET_DEBUG_MARKS
        if (dbg) D3_MARK(1);
#endif
        const int ntiles = (M + BOX_PTS - 1) / BOX_PTS;
        for (int r0 = 0; r0 < ntiles; r0 += 64) {
            const float db = round_box_d2(B.box2, P, r0, lane, ntiles, qx, qy, qz) * BOX_MARGIN;
            unsigned long long todo = __ballot(db < TOL2 && BOX_PTS * (r0 + lane) < q);   // (a tile behind the query holds no earlier point)
#ifdef RDET_DEBUG_MARKS
            if (dbg && r0 == 0) D3_MARK(2);
            dbg_tiles += __popcll(todo);
#endif
            while (todo) {
#ifdef RDET_DEBUG_MARKS
                ++dbg_batches; dbg_t = wall_clock64();
#endif
                float cx[CC_AHEAD], cy[CC_AHEAD], cz[CC_AHEAD];
                int top[CC_AHEAD];
                bool ok[CC_AHEAD];
#pragma unroll
                for (int u = 0; u < CC_AHEAD; ++u) {                        // four steps' loads in flight
                    int ta = -1, tb = -1;
                    if (todo) { ta = __ffsll((long long)todo) - 1; todo &= todo - 1; }
                    if (todo) { tb = __ffsll((long long)todo) - 1; todo &= todo - 1; }
                    const int tile = (lane < 32) ? ta : tb;
                    const int j = BOX_PTS * (r0 + tile) + (lane & 31);
                    // every edge is handled by its later end (j < q also keeps j < M)
                    ok[u] = tile >= 0 && j < q;
                    const int jj = ok[u] ? j : 0;
                    cx[u] = X[jj]; cy[u] = Y[jj]; cz[u] = Z[jj];
                    top[u] = uf_ld(&parent[B.perm[jj]]);       // (with the coordinates, not behind the test; -1: an outlier)
                }
                // the tops of the neighbours' chains, all lanes and steps together (a lane without a neighbour rides along on the query's own top)
#pragma unroll
                for (int u = 0; u < CC_AHEAD; ++u)
                    if (!(ok[u] && top[u] >= 0 && d2f(qx, qy, qz, cx[u], cy[u], cz[u]) < TOL2)) top[u] = rs;
#ifdef RDET_DEBUG_MARKS
                { const unsigned long long t = wall_clock64(); dbg_tl += t - dbg_t; dbg_t = t; }
#endif
                bool moving = true;
                while (__ballot(moving)) {
#ifdef RDET_DEBUG_MARKS
                    ++dbg_hops;
#endif
                    int p[CC_AHEAD];
#pragma unroll
                    for (int u = 0; u < CC_AHEAD; ++u) p[u] = uf_ld(&parent[top[u]]);
                    moving = false;
#pragma unroll
                    for (int u = 0; u < CC_AHEAD; ++u) { moving |= p[u] != top[u]; top[u] = p[u]; }
                }
#ifdef RDET_DEBUG_MARKS
                { const unsigned long long t = wall_clock64(); dbg_tc += t - dbg_t; dbg_t = t; }
#endif
                // one union per DISTINCT top that is not the query's: the k-th distinct one is lane k's, all of them at once (64 per round)
                while (true) {
                    int mine = -1, nd = 0;
#pragma unroll
                    for (int u = 0; u < CC_AHEAD; ++u) {
                        unsigned long long need = __ballot(top[u] != rs && top[u] != ri);
                        while (need && nd < 64) {
                            const int c = __builtin_amdgcn_readlane(top[u], __ffsll((long long)need) - 1);
                            need &= ~__ballot(top[u] == c);
#pragma unroll
                            for (int v = 0; v < CC_AHEAD; ++v) top[v] = (top[v] == c) ? rs : top[v];   // (handled, in whichever step it turns up)
                            if (lane == nd) mine = c;
                            ++nd;
                        }
                    }
                    if (nd == 0) break;
#ifdef RDET_DEBUG_MARKS
                    ++dbg_rounds;
#endif
                    int a = ri;
                    if (mine >= 0) a = uf_union(parent, ri, mine);
                    for (int off = 32; off > 0; off >>= 1) a = min(a, __shfl_xor(a, off, 64));
                    ri = a;                                                 // (the smallest root any lane has seen: a hint, like rs)
                }
#ifdef RDET_DEBUG_MARKS
                { const unsigned long long t = wall_clock64(); dbg_tu += t - dbg_t; dbg_t = t; }
#endif
            }
        }
#ifdef RDET_DEBUG_MARKS
        if (dbg) {
            D3_MARK(3);
            D3_NOTE(4, dbg_tl); D3_NOTE(5, dbg_tc);
            D3_NOTE(6, ((unsigned long long)dbg_tiles << 32) | (unsigned)(dbg_batches * 10000 + dbg_hops * 100 + dbg_rounds));
            D3_NOTE(7, dbg_tu);
        }
#endif
    }
}

#undef D3_KERNEL
#define D3_KERNEL 0
// ---- sizes, gate, order, centroids ------------------------------------------------------------------
// k3_finish_a (thread = sorted position, over the CUs): final roots, component sizes, the stretch of the sorted copy
// each component lives in, and the list of roots.  One atomic group per (wave, component) instead of one per point:
// neighbours in the spatial order mostly share their component.
__global__ __launch_bounds__(256) void k3_finish_a(Det3dBufs B, int N)
{
    const int M = B.ctl->M;
    const int s = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    const int node0 = B.perm[s < N ? s : 0];                      // (beside M, not behind it)
    if ((int)blockIdx.x * 256 >= M) return;
    int r = -1, node = -1;
    if (s < M) {
        node = node0;
        r = B.label[node];
        if (r >= 0) {
            while (true) { const int p = uf_ld(&B.label[r]); if (p == r) break; r = p; }   // parents are final: no union runs any more
        }
        reinterpret_cast<int *>(B.dist_s)[s] = r;                // the SOR distances are dead: final roots live there, by sorted position
    }
    const unsigned long long isroot = __ballot(r >= 0 && r == node);
    if (isroot) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&B.ctl->nroots, __popcll(isroot));
        base = __shfl(base, 0, 64);
        if (r >= 0 && r == node) B.roots[base + __popcll(isroot & ((1ull << lane) - 1))] = r;
    }
    unsigned long long todo = __ballot(r >= 0);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        const int r0 = __shfl(r, src, 64);
        const unsigned long long grp = __ballot(r == r0);
        if (lane == src) {
            const int sbase = blockIdx.x * 256 + (threadIdx.x & ~63u);
            atomicAdd(&B.cnt[r0], __popcll(grp));
            atomicMin(&B.first[r0], sbase + src);
            atomicMax(&B.last[r0], sbase + 63 - __clzll((long long)grp));
        }
        todo &= ~grp;
    }
}
// k3_clusters: EVERY workgroup gates and ranks the components for itself (a few hundred roots: cheaper than a launch in
// between), then each of its four waves takes one accepted component: its members out of the stretch of the sorted copy
// it lives in, brought into ARRIVAL order (the float32 sums of compute3DCentroid run in index order, :94), summed,
// divided, moved to base_link and published.  Order: size descending, then first member (= root) ascending.
template <int NS>
__device__ static inline void arrival_places(const int *ids, int size, int lane, int (&place)[3])
{
    // a member's place = the number of members with a smaller node id (ids four at a time: the list is padded with ids no member is above)
    int mine[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { mine[s] = (lane + 64 * s < size) ? ids[lane + 64 * s] : 0; place[s] = 0; }
#pragma unroll 4
    for (int q = 0; q < size; q += 4) {                           // (unrolled: the reads of four rounds in flight, not one LDS round trip per round)
        const int4 o = *(const int4 *)&ids[q];
#pragma unroll
        for (int s = 0; s < NS; ++s) place[s] += (o.x < mine[s]) + (o.y < mine[s]) + (o.z < mine[s]) + (o.w < mine[s]);
    }
}
__global__ __launch_bounds__(256) void k3_clusters(Det3dBufs B, int max_centers, float sx, float sy, float cs, float sn)
{
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) unsigned long long s_key[RDET_MAX_CENTERS];
    __shared__ int s_root[RDET_MAX_CENTERS], s_size[RDET_MAX_CENTERS], s_first[RDET_MAX_CENTERS], s_lastp[RDET_MAX_CENTERS];
    __shared__ int s_byrank[RDET_MAX_CENTERS];
    __shared__ int s_n, s_err;
    __shared__ __attribute__((aligned(16))) int m_id[4][MAX_SZ + 4];
    __shared__ __attribute__((aligned(16))) float m_x[4][MAX_SZ], m_y[4][MAX_SZ], o_x[4][MAX_SZ], o_y[4][MAX_SZ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    D3_MARK(0);
    const int nroots = B.ctl->nroots;
    if (tid == 0) { s_n = 0; s_err = 0; }
    s_key[tid] = ~0ull;                                                           // (no real key is above it)
    __syncthreads();
    for (int k = tid; k < nroots; k += 256) {
        const int r = B.roots[k], c = B.cnt[r], f = B.first[r], l = B.last[r];   // (extents with the size, not behind the ranking)
        if (c >= MIN_SZ && c <= MAX_SZ) {                                         // :70-71
            const int pos = atomicAdd(&s_n, 1);
            if (pos < RDET_MAX_CENTERS) {
                s_root[pos] = r; s_size[pos] = c; s_first[pos] = f; s_lastp[pos] = l;
                s_key[pos] = ((unsigned long long)(unsigned)(MAX_SZ - c) << 32) | (unsigned)r;
            } else s_err = RDET_ERR_CAPACITY;
        }
    }
    __syncthreads();
    D3_MARK(1);
    int n = min(s_n, RDET_MAX_CENTERS);
    int err = s_err;
    if (n > max_centers) { err = RDET_ERR_BUFFER; n = 0; }
    if (B.ctl->retry) { err = D3_RETRY; n = 0; }
    // rank = the number of keys below one's own ((MAX_SZ - size, root): size descending, root ascending); four threads per entry, two
    // keys per LDS read (round 4: one thread per entry, size and root apart: 3 us for 76 entries)
    for (int e0 = 0; e0 < n; e0 += 64) {
        const int e = e0 + (tid >> 2), sub = tid & 3;
        const unsigned long long me = s_key[e];
        int rank = 0;
#pragma unroll 4
        for (int k = 2 * sub; k < n; k += 8) {
            const ulonglong2 kk = *(const ulonglong2 *)&s_key[k];
            rank += (kk.x < me) + (kk.y < me);
        }
        rank += __shfl_xor(rank, 1, 64); rank += __shfl_xor(rank, 2, 64);
        if (sub == 0 && e < n) s_byrank[rank] = e;
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0) {
        B.ctl->K = n; B.ctl->err = err;
        d3_host_store16(&B.hout->head, (unsigned)n, (unsigned)err, (unsigned)B.ctl->m_true, (unsigned)B.seq);   // the centres follow, each with its own tag
    }
    // the next cloud's grid: the box of this cloud's inliers (= of the tiles' tight boxes), a little wider, at least 1/8 m per cell -- by the LAST
    // workgroup, the one least likely to have a component to sum (round 4: the first one, which has the largest)
    if (blockIdx.x == gridDim.x - 1 && wave == 3) {
        const int ntiles = (B.ctl->M + BOX_PTS - 1) / BOX_PTS;
        float x0 = INFINITY, y0 = INFINITY, x1 = -INFINITY, y1 = -INFINITY;
        for (int t = lane; t < ntiles; t += 64) {
            const float4 lo = *(const float4 *)(B.box2 + 8 * t), hi = *(const float4 *)(B.box2 + 8 * t + 4);
            x0 = fminf(x0, lo.x); y0 = fminf(y0, lo.y); x1 = fmaxf(x1, lo.w); y1 = fmaxf(y1, hi.x);
        }
        for (int off = 32; off > 0; off >>= 1) {
            x0 = fminf(x0, __shfl_xor(x0, off, 64)); y0 = fminf(y0, __shfl_xor(y0, off, 64));
            x1 = fmaxf(x1, __shfl_xor(x1, off, 64)); y1 = fmaxf(y1, __shfl_xor(y1, off, 64));
        }
        if (lane == 0 && x0 <= x1 && y0 <= y1 && fabsf(x0) < 1e30f && fabsf(x1) < 1e30f && fabsf(y0) < 1e30f && fabsf(y1) < 1e30f) {
            const float ext = fmaxf(fmaxf(x1 - x0, y1 - y0) * 1.05f, 0.125f * GRID_G);
            B.ctl->gx0 = 0.5f * (x0 + x1) - 0.5f * ext; B.ctl->gy0 = 0.5f * (y0 + y1) - 0.5f * ext; B.ctl->ginv = (float)GRID_G / ext;
        }
    }
    D3_MARK(2);
    const int rank = blockIdx.x * 4 + wave;
    if (rank >= n) return;
    const int e = s_byrank[rank], root = s_root[e], size = s_size[e];
    const int *slabel = reinterpret_cast<const int *>(B.dist_s);
    const int first = s_first[e], last = s_lastp[e];
    int have = 0;
    // a component's members sit close together in the sorted copy -- except when it straddles a major Morton boundary
    // (stretches of a few thousand positions occur): sixteen chunks' labels per round trip, then the members' data
    for (int b0 = first; b0 <= last && have < size; b0 += 1024) {
        int sl[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int s = b0 + 64 * u + lane; sl[u] = (s <= last) ? slabel[s] : -2; }
        int id[16];
        float mx[16], my[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int s = b0 + 64 * u + lane;
            if (sl[u] == root) { id[u] = B.perm[s]; mx[u] = B.s1[s]; my[u] = B.s1[B.cap + s]; }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const bool mem = sl[u] == root;
            const unsigned long long mask = __ballot(mem);
            if (mem) {
                const int k = have + __popcll(mask & ((1ull << lane) - 1));
                m_id[wave][k] = id[u]; m_x[wave][k] = mx[u]; m_y[wave][k] = my[u];
            }
            have += __popcll(mask);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    D3_MARK(3);
#ifdef RDET_DEBUG_MARKS
    if (threadIdx.x == 0 && blockIdx.x < 2048) { d3_marks[blockIdx.x][6] = (unsigned long long)(last - first + 1); d3_marks[blockIdx.x][7] = (unsigned long long)size; }
#endif
    if (lane < 4) m_id[wave][size + lane] = 0x7fffffff;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        int place[3] = {0, 0, 0};
        if (size <= 64) arrival_places<1>(m_id[wave], size, lane, place);
        else if (size <= 128) arrival_places<2>(m_id[wave], size, lane, place);
        else arrival_places<3>(m_id[wave], size, lane, place);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int k = lane + 64 * s;
            if (k < size) { o_x[wave][place[s]] = m_x[wave][k]; o_y[wave][place[s]] = m_y[wave][k]; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float cx = 0.f, cy = 0.f;
    int q = 0;
#pragma unroll 4
    for (; q + 4 <= size; q += 4) {
        const float4 vx = *(const float4 *)&o_x[wave][q], vy = *(const float4 *)&o_y[wave][q];
        cx += vx.x; cx += vx.y; cx += vx.z; cx += vx.w;
        cy += vy.x; cy += vy.y; cy += vy.z; cy += vy.w;
    }
    for (; q < size; ++q) { cx += o_x[wave][q]; cy += o_y[wave][q]; }
    if (lane == 0) {
        const float sz = (float)size;
        cx /= sz; cy /= sz;
        const float ox = (cs * cx + (-sn) * cy) + sx, oy = (sn * cx + cs * cy) + sy;   // :96 Project2D(s2b).cast<float>() * p
        d3_host_store16(&B.hout->centers[rank], __float_as_uint(ox), __float_as_uint(oy), (unsigned)B.seq, 0u);
    }
    D3_MARK(4);
}
#undef D3_KERNEL

// ---- the short back end (round 6): k3_finish_a + k3_clusters in ONE launch for clouds of at most MFAST survivors ---------------------
// Every workgroup takes the whole forest into LDS for itself -- parents by node (one trip to memory), then roots by chasing in LDS (a hop is
// an LDS access, not a trip to L2), sizes by run-length counting (consecutive nodes mostly share their component), the members' (x, y) by
// node -- gates and ranks the components like k3_clusters, and each of its four waves sums one component: its members are the nodes whose
// root it is, found IN NODE ORDER = arrival order by a sweep over the LDS table, so the float32 sums of compute3DCentroid (:94) need no
// re-ordering (k3_clusters gathers from the sorted copy and sorts by node id: O(size^2)).  No sizes, extents or root lists in memory.
#define D3_KERNEL 5
constexpr int FC_T = 1024, FC_W = FC_T / 64;                                      // threads / waves = components per workgroup
__global__ __launch_bounds__(FC_T) void k3f_clusters(Det3dBufs B, int max_centers, float sx, float sy, float cs, float sn)
{
#pragma clang fp contract(off)
    __shared__ int par[MFAST];                                                    // by node: parent, then root (-1: SOR's outliers)
    __shared__ int csz[MFAST];                                                    // by root: component size
    __shared__ float nx[MFAST], ny[MFAST];                                        // by node
    __shared__ __attribute__((aligned(16))) unsigned long long s_key[RDET_MAX_CENTERS];
    __shared__ int s_root[RDET_MAX_CENTERS], s_size[RDET_MAX_CENTERS], s_byrank[RDET_MAX_CENTERS];
    __shared__ int s_n, s_err;
    __shared__ __attribute__((aligned(16))) float m_x[FC_W][MAX_SZ], m_y[FC_W][MAX_SZ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    D3_MARK(0);
    const int M = B.ctl->M;
    if (tid == 0) { s_n = 0; s_err = 0; }
    if (tid < RDET_MAX_CENTERS) s_key[tid] = ~0ull;                                                           // (no real key is above it)
    // one trip: parents by node; coordinates and node numbers by sorted position (eight of each per thread in flight)
    for (int b0 = 0; b0 < M; b0 += FC_T * 8) {
        int lb[8], pn[8];
        float px[8], py[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = b0 + tid + FC_T * k;
            const bool v = i < M;
            lb[k] = v ? B.label[i] : -1;
            pn[k] = v ? B.perm[i] : 0; px[k] = v ? B.s1[i] : 0.f; py[k] = v ? B.s1[B.cap + i] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = b0 + tid + FC_T * k;
            if (i < M) { par[i] = lb[k]; csz[i] = 0; nx[pn[k]] = px[k]; ny[pn[k]] = py[k]; }
        }
    }
    __syncthreads();
    D3_MARK(1);
    // roots + sizes: a thread takes four consecutive nodes (started at an offset that keeps the lanes off each other's banks), chases them
    // side by side, and adds a run of equal roots to that root's size with one atomic.  (Sixteen nodes per thread, the first version,
    // left three quarters of the workgroup idle at M = 3 k and took 6 us: a wave's chase is a chain of LDS round trips.)
    for (int blk = tid; 4 * blk < M; blk += FC_T) {
        int r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int n = 4 * blk + ((j + (blk >> 4)) & 3); r[j] = (n < M) ? par[n] : -1; }
        bool moving = true;
        while (moving) {
            moving = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int p = (r[j] >= 0) ? par[r[j]] : -1; moving |= p != r[j]; r[j] = p; }
        }
        int run_root = -1, run = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = 4 * blk + ((j + (blk >> 4)) & 3);
            if (n < M && r[j] >= 0) par[n] = r[j];                               // (a root is an ancestor: whoever still chases through n stays right)
            if (r[j] != run_root) { if (run_root >= 0) atomicAdd(&csz[run_root], run); run_root = r[j]; run = 0; }
            ++run;
        }
        if (run_root >= 0) atomicAdd(&csz[run_root], run);
    }
    __syncthreads();
    D3_MARK(2);
    for (int n = tid; n < M; n += FC_T) {
        const int c = (par[n] == n) ? csz[n] : 0;
        if (c >= MIN_SZ && c <= MAX_SZ) {                                         // :70-71
            const int pos = atomicAdd(&s_n, 1);
            if (pos < RDET_MAX_CENTERS) {
                s_root[pos] = n; s_size[pos] = c;
                s_key[pos] = ((unsigned long long)(unsigned)(MAX_SZ - c) << 32) | (unsigned)n;
            } else s_err = RDET_ERR_CAPACITY;
        }
    }
    __syncthreads();
    D3_MARK(3);
    int n = min(s_n, RDET_MAX_CENTERS);
    int err = s_err;
    if (n > max_centers) { err = RDET_ERR_BUFFER; n = 0; }
    if (B.ctl->retry) { err = D3_RETRY; n = 0; }
    // rank = the number of keys below one's own ((MAX_SZ - size, root): size descending, root ascending); four threads per entry
    for (int e0 = 0; e0 < n; e0 += FC_T / 4) {
        const int e = e0 + (tid >> 2), sub = tid & 3;
        if (e >= RDET_MAX_CENTERS) break;
        const unsigned long long me = s_key[e];
        int rank = 0;
#pragma unroll 4
        for (int k = 2 * sub; k < n; k += 8) {
            const ulonglong2 kk = *(const ulonglong2 *)&s_key[k];
            rank += (kk.x < me) + (kk.y < me);
        }
        rank += __shfl_xor(rank, 1, 64); rank += __shfl_xor(rank, 2, 64);
        if (sub == 0 && e < n) s_byrank[rank] = e;
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0) {
        B.ctl->K = n; B.ctl->err = err;
        d3_host_store16(&B.hout->head, (unsigned)n, (unsigned)err, (unsigned)B.ctl->m_true, (unsigned)B.seq);   // the centres follow, each with its own tag
    }
    // the next cloud's grid: as k3_clusters
    D3_MARK(4);
    if (blockIdx.x == gridDim.x - 1 && wave == FC_W - 1) {
        const int ntiles = (M + BOX_PTS - 1) / BOX_PTS;
        float x0 = INFINITY, y0 = INFINITY, x1 = -INFINITY, y1 = -INFINITY;
        for (int t = lane; t < ntiles; t += 64) {
            const float4 lo = *(const float4 *)(B.box2 + 8 * t), hi = *(const float4 *)(B.box2 + 8 * t + 4);
            x0 = fminf(x0, lo.x); y0 = fminf(y0, lo.y); x1 = fmaxf(x1, lo.w); y1 = fmaxf(y1, hi.x);
        }
        for (int off = 32; off > 0; off >>= 1) {
            x0 = fminf(x0, __shfl_xor(x0, off, 64)); y0 = fminf(y0, __shfl_xor(y0, off, 64));
            x1 = fmaxf(x1, __shfl_xor(x1, off, 64)); y1 = fmaxf(y1, __shfl_xor(y1, off, 64));
        }
        if (lane == 0 && x0 <= x1 && y0 <= y1 && fabsf(x0) < 1e30f && fabsf(x1) < 1e30f && fabsf(y0) < 1e30f && fabsf(y1) < 1e30f) {
            const float ext = fmaxf(fmaxf(x1 - x0, y1 - y0) * 1.05f, 0.125f * GRID_G);
            B.ctl->gx0 = 0.5f * (x0 + x1) - 0.5f * ext; B.ctl->gy0 = 0.5f * (y0 + y1) - 0.5f * ext; B.ctl->ginv = (float)GRID_G / ext;
        }
    }
    const int rank = blockIdx.x * FC_W + wave;
    if (rank >= n) return;
    const int e = s_byrank[rank], root = s_root[e], size = s_size[e];
    // the members in node order: the nodes whose root this is (none is below the root: a component's label is its smallest node)
    int have = 0;
    for (int n0 = root & ~63; n0 < M && have < size; n0 += 512) {                 // (eight table reads in flight; a block without a member costs a compare)
        int pr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int nn = n0 + 64 * u + lane; pr[u] = (nn < M) ? par[nn] : -1; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int nn = n0 + 64 * u + lane;
            const bool mem = pr[u] == root;
            const unsigned long long mask = __ballot(mem);
            if (mask == 0ull) continue;                                           // (reading every block's coordinates up front, predicated, was slower: 2.8 -> 4.3 us for 112 members)
            if (mem) {
                const int k = have + __popcll(mask & ((1ull << lane) - 1));
                m_x[wave][k] = nx[nn]; m_y[wave][k] = ny[nn];
            }
            have += __popcll(mask);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    D3_MARK(5);
    float cx = 0.f, cy = 0.f;
    int q = 0;
#pragma unroll 4
    for (; q + 4 <= size; q += 4) {
        const float4 vx = *(const float4 *)&m_x[wave][q], vy = *(const float4 *)&m_y[wave][q];
        cx += vx.x; cx += vx.y; cx += vx.z; cx += vx.w;
        cy += vy.x; cy += vy.y; cy += vy.z; cy += vy.w;
    }
    for (; q < size; ++q) { cx += m_x[wave][q]; cy += m_y[wave][q]; }
    if (lane == 0) {
        const float sz = (float)size;
        cx /= sz; cy /= sz;
        const float ox = (cs * cx + (-sn) * cy) + sx, oy = (sn * cx + cs * cy) + sy;   // :96 Project2D(s2b).cast<float>() * p
        d3_host_store16(&B.hout->centers[rank], __float_as_uint(ox), __float_as_uint(oy), (unsigned)B.seq, 0u);
    }
    D3_MARK(6);
}
#undef D3_KERNEL

}  // namespace

struct rdet3d {
    rdet3d_options opt;
    double s2b[3];
    int max_points, device;
    // TWO clouds may be on their way (rdet3d_submit / rdet3d_collect), and their chains of kernels run SIDE BY SIDE: a slot is a complete
    // detector -- input buffer, every intermediate array, control block, result slots and a stream of its own (round 5 shared everything
    // between the input and the results, so cloud k + 1's chain queued behind cloud k's; most of these kernels are a few dozen workgroups
    // of latency chain and leave the chip nearly empty).  The slots alternate; a slot's next cloud follows its previous one on its stream,
    // and inherits its sorting grid.
    struct Slot {
        hipStream_t stream;
        float *d_p1, *d_s1, *d_dist, *d_dist_s, *d_box, *d_box2;
        int *d_label, *d_cnt, *d_first, *d_last, *d_roots, *d_perm, *d_hist, *d_cursor;
        Det3dCtl *d_ctl;
        bool in_flight;                // a call returned before this slot's kernels had published everything
        float *d_xyzi;                 // the cloud on the device
        bool xyzi_in_vram;             // ... in fine-grained device memory the host writes through the PCIe BAR (else: pinned staging + copy)
        float *h_stage;
        Det3dHostOut *h_out, *dv_out;  // pinned + mapped: polled result slots (host / device view)
        bool busy;                     // submitted, not collected
        bool fast;                     // its chain starts with the short front end
        int seq, max_centers;
        int N;                         // points of the submitted cloud (a chain that has to be launched again: D3_RETRY)
        double stamp;
    } slot[2];
    int n_out;                         // clouds submitted and not collected (0 .. 2); the older one is slot[(next + 2 - n_out) & 1]
    int next;                          // the slot the next submit takes
    int seq;
    int m_hint;                        // the previous cloud's survivors of the gate (k3_knn: m_hint; which front end the next cloud gets)
    int path_mode;                     // 0: by m_hint; 1: always the long chain; 2: always try the short front end (rdet3d_debug_set_path)
    unsigned long long n_short, n_retry;   // clouds sent through the short front end / of those, sent again through the long one
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

// one cloud's chain of kernels onto the handle's stream (the cloud is in sl.d_xyzi): the short front end when the previous cloud's survivors
// would have fitted it, else the long one
static int d3_launch(rdet3d_t *h, rdet3d::Slot &sl)
{
    const int N = sl.N, max_centers = sl.max_centers;
    Det3dBufs B;
    B.xyzi = sl.d_xyzi; B.p1 = sl.d_p1; B.s1 = sl.d_s1; B.dist_s = sl.d_dist_s; B.perm = sl.d_perm; B.box = sl.d_box; B.box2 = sl.d_box2; B.hist = sl.d_hist; B.cursor = sl.d_cursor;
    B.dist = sl.d_dist; B.label = sl.d_label; B.cnt = sl.d_cnt; B.first = sl.d_first; B.last = sl.d_last; B.roots = sl.d_roots;
    B.ctl = sl.d_ctl; B.cap = h->max_points; B.cap1 = (h->max_points + 1023) & ~1023;
    B.hout = sl.dv_out; B.seq = sl.seq = ++h->seq;
    const int ftiles = (N + 1023) / 1024, b256 = (N + 255) / 256, ntiles_ub = (N + BOX_PTS - 1) / BOX_PTS;
    sl.fast = ftiles <= 1024 && (h->path_mode == 2 || (h->path_mode == 0 && h->m_hint <= MFAST - MFAST / 10));
    if (sl.fast) {
        ++h->n_short;
#ifdef D3_TWO_LAUNCH_FRONT
        hipLaunchKernelGGL(k3f_gate, dim3(ftiles), dim3(1024), 0, sl.stream, B, N, h->opt.intensity_min);
        hipLaunchKernelGGL(k3f_sort, dim3(1), dim3(1024), 0, sl.stream, B, N, ftiles);
#else
        hipLaunchKernelGGL(k3f_front, dim3(ftiles), dim3(1024), 0, sl.stream, B, N, h->opt.intensity_min, ftiles);
#endif
    } else {
        hipLaunchKernelGGL(k3_filter_count, dim3(ftiles), dim3(1024), 0, sl.stream, B, N, h->opt.intensity_min);
        hipLaunchKernelGGL(k3_filter_write, dim3(ftiles + GRID_CELLS / 1024), dim3(1024), 0, sl.stream, B, N, h->opt.intensity_min, ftiles);
        hipLaunchKernelGGL(k3_scatter, dim3(ftiles), dim3(1024), 0, sl.stream, B, N);
        hipLaunchKernelGGL(k3_boxes, dim3(b256), dim3(256), 0, sl.stream, B, N);
    }
    const int qblocks = (N + QW - 1) / QW < Q_GRID ? (N + QW - 1) / QW : Q_GRID;   // a wave per query, dealt round-robin: M <= N stays on the device
    hipLaunchKernelGGL(k3_knn, dim3(qblocks), dim3(64 * QW), 0, sl.stream, B, N, ntiles_ub, h->m_hint);
    hipLaunchKernelGGL(k3_cc_min, dim3(qblocks), dim3(64 * QW), 0, sl.stream, B, N, ntiles_ub, h->m_hint);
    hipLaunchKernelGGL(k3_cc_link, dim3(qblocks), dim3(64 * QW), 0, sl.stream, B, N, ntiles_ub, h->m_hint);
    const float sa = (float)h->s2b[2];
    if (sl.fast) {                                                     // (M <= MFAST, or M = 0 and D3_RETRY)
        hipLaunchKernelGGL(k3f_clusters, dim3(RDET_MAX_CENTERS / FC_W), dim3(FC_T), 0, sl.stream, B, max_centers < RDET_MAX_CENTERS ? max_centers : RDET_MAX_CENTERS,
                           (float)h->s2b[0], (float)h->s2b[1], cosf(sa), sinf(sa));
    } else {
        hipLaunchKernelGGL(k3_finish_a, dim3(b256), dim3(256), 0, sl.stream, B, N);
        hipLaunchKernelGGL(k3_clusters, dim3(RDET_MAX_CENTERS / 4), dim3(256), 0, sl.stream, B, max_centers < RDET_MAX_CENTERS ? max_centers : RDET_MAX_CENTERS,
                           (float)h->s2b[0], (float)h->s2b[1], cosf(sa), sinf(sa));
    }
    DET3_TRY(h, hipGetLastError());
    return RDET_OK;
}

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
        for (auto &sl : h->slot) {
            DET3_TRY(h, hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
            sl.d_xyzi = (float *)host_visible::alloc(16 * np);
            if (sl.d_xyzi) sl.xyzi_in_vram = true;
            else {
                sl.xyzi_in_vram = false;
                DET3_TRY(h, hipMalloc(&sl.d_xyzi, 16 * np));
                DET3_TRY(h, hipHostMalloc(&sl.h_stage, 16 * np));
            }
            // + 1024 floats: the sweeps ask for the aligned 64 points around a query before they know M (up to 63 past the end)
            DET3_TRY(h, hipMalloc(&sl.d_p1, 12 * (np + 1024) + 4096)); DET3_TRY(h, hipMemset(sl.d_p1, 0, 12 * (np + 1024) + 4096));   // (+ 1024: k3f_front's whole tiles)
            DET3_TRY(h, hipMalloc(&sl.d_s1, 12 * np + 4096)); DET3_TRY(h, hipMemset(sl.d_s1, 0, 12 * np + 4096));
            DET3_TRY(h, hipMalloc(&sl.d_perm, 4 * np + 4096)); DET3_TRY(h, hipMemset(sl.d_perm, 0, 4 * np + 4096));
            DET3_TRY(h, hipMalloc(&sl.d_box, 32 * (np / BOX_PTS + 2)));
            DET3_TRY(h, hipMalloc(&sl.d_box2, 32 * (np / BOX_PTS + 2)));
            DET3_TRY(h, hipMalloc(&sl.d_hist, 4 * GRID_CELLS)); DET3_TRY(h, hipMemset(sl.d_hist, 0, 4 * GRID_CELLS));
            DET3_TRY(h, hipMalloc(&sl.d_cursor, 4 * GRID_CELLS));
            DET3_TRY(h, hipMalloc(&sl.d_dist, 4 * np + 4096)); DET3_TRY(h, hipMemset(sl.d_dist, 0, 4 * np + 4096));
            DET3_TRY(h, hipMalloc(&sl.d_dist_s, 4 * np + 4096)); DET3_TRY(h, hipMemset(sl.d_dist_s, 0, 4 * np + 4096));
            DET3_TRY(h, hipMalloc(&sl.d_label, 4 * np));
            DET3_TRY(h, hipMalloc(&sl.d_cnt, 4 * np));
            DET3_TRY(h, hipMalloc(&sl.d_last, 4 * np));
            DET3_TRY(h, hipMalloc(&sl.d_first, 4 * np));
            DET3_TRY(h, hipMalloc(&sl.d_roots, 4 * np));
            DET3_TRY(h, hipMalloc(&sl.d_ctl, sizeof(Det3dCtl)));
            {   // a slot's first cloud is sorted on a 64 m x 64 m grid around the sensor; every later one on its predecessor's box
                Det3dCtl c0;
                std::memset(&c0, 0, sizeof(c0));
                c0.gx0 = c0.gy0 = -32.f; c0.ginv = (float)GRID_G / 64.f;
                DET3_TRY(h, hipMemcpy(sl.d_ctl, &c0, sizeof(c0), hipMemcpyHostToDevice));
            }
        }
        for (auto &sl : h->slot) {
            DET3_TRY(h, hipHostMalloc(&sl.h_out, sizeof(Det3dHostOut), hipHostMallocMapped | hipHostMallocCoherent));
            std::memset(sl.h_out, 0, sizeof(Det3dHostOut));
            void *dv = nullptr;
            DET3_TRY(h, hipHostGetDevicePointer(&dv, sl.h_out, 0)); sl.dv_out = (Det3dHostOut *)dv;
        }
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
    for (auto &sl : h->slot) {
        if (sl.stream) (void)hipStreamSynchronize(sl.stream);
        if (sl.d_xyzi) (void)hipFree(sl.d_xyzi);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.h_stage) (void)hipHostFree(sl.h_stage);
        void *ptrs[] = {sl.d_p1, sl.d_s1, sl.d_dist, sl.d_dist_s, sl.d_box, sl.d_box2, sl.d_label, sl.d_cnt, sl.d_first, sl.d_last, sl.d_roots, sl.d_perm, sl.d_hist, sl.d_cursor, sl.d_ctl};
        for (void *p : ptrs) (void)hipFree(p);
        if (sl.stream) (void)hipStreamDestroy(sl.stream);
    }
    delete h;
}

// The two halves of HandlePointCloud.  rdet3d_submit: the cloud into the device's memory and its chain of launches, no waiting; rdet3d_collect:
// the OLDEST submitted cloud's centres, polled from its result slots.  Up to two clouds may be submitted and not collected: the host copies
// and enqueues cloud k + 1 while the device is still on cloud k -- on the other slot's stream and arrays, so the two chains share the chip.
int rdet3d_submit(rdet3d_t *h, double stamp, const float *xyzi, int N, int max_centers)
{
    if (!h || N < 0 || (N > 0 && !xyzi) || max_centers < 0) return RDET_ERR_INVALID;
    if (N > h->max_points) return RDET_ERR_CAPACITY;
    if (h->n_out >= 2) return RDET_ERR_INVALID;                        // (collect first: two clouds are on their way)
    rdet3d::Slot &sl = h->slot[h->next];
    sl.stamp = stamp; sl.max_centers = max_centers; sl.seq = 0;       // (seq 0: an empty cloud, nothing to wait for)
    if (N > 0) {
#ifdef RDET_DEBUG_MARKS
        const auto dbg_t0 = std::chrono::steady_clock::now();
        auto dbg_us = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - dbg_t0).count(); };
        double dbg_t[3] = {0, 0, 0};
#endif
        DET3_TRY(h, hipSetDevice(h->device));
        // Every result slot of a cloud is written by the LAST kernel of its chain, which starts when all others have ended: a cloud that has
        // been collected reads its input no more, and this slot's previous cloud has been collected (n_out < 2).  Only a call that gave up
        // waiting leaves kernels in flight.  (hipStreamSynchronize on an idle stream costs 15 us.)
        if (sl.in_flight) { DET3_TRY(h, hipStreamSynchronize(sl.stream)); sl.in_flight = false; }
#ifdef RDET_DEBUG_MARKS
        dbg_t[0] = dbg_us();
#endif
        if (sl.xyzi_in_vram) {                                         // the cloud goes straight into device memory: posted writes, no copy engine
            std::memcpy(sl.d_xyzi, xyzi, sizeof(float) * 4 * (size_t)N);
            __atomic_thread_fence(__ATOMIC_SEQ_CST);
        } else {
            std::memcpy(sl.h_stage, xyzi, sizeof(float) * 4 * (size_t)N);
            DET3_TRY(h, hipMemcpyAsync(sl.d_xyzi, sl.h_stage, sizeof(float) * 4 * (size_t)N, hipMemcpyHostToDevice, sl.stream));
        }
#ifdef RDET_DEBUG_MARKS
        dbg_t[1] = dbg_us();
#endif
        sl.N = N;
        { const int rc = d3_launch(h, sl); if (rc != RDET_OK) return rc; }
#ifdef RDET_DEBUG_MARKS
        dbg_t[2] = dbg_us();
        if (getenv("RDET3_HOST_MARKS")) std::fprintf(stderr, "rdet3d host us: synced %.1f cloud written %.1f launched %.1f\n", dbg_t[0], dbg_t[1], dbg_t[2]);
#endif
    }
    sl.busy = true;
    h->next ^= 1;
    ++h->n_out;
    return RDET_OK;
}

int rdet3d_collect(rdet3d_t *h, float *centers_xy, int max_centers, int *K, double *obs_time)
{
    if (!h || !K || max_centers < 0 || (max_centers > 0 && !centers_xy)) return RDET_ERR_INVALID;
    *K = 0;
    if (h->n_out == 0) return RDET_ERR_INVALID;                        // nothing was submitted
    rdet3d::Slot &sl = h->slot[(h->next + 2 - h->n_out) & 1];          // the older of the clouds on their way
    // (room for what the submit promised -- checked BEFORE the slot is given up: the caller can come again with a larger buffer, the cloud's
    // result is not lost, and nobody writes the slot's input buffer under a chain that has not been waited for)
    if (sl.seq != 0 && max_centers < sl.max_centers && max_centers < RDET_MAX_CENTERS) return RDET_ERR_BUFFER;
    --h->n_out;
    sl.busy = false;
    if (obs_time) *obs_time = sl.stamp;                               // :16
    if (sl.seq == 0) return RDET_OK;                                  // an empty cloud
    // poll the head, then each centre's own tag (k3_clusters)
    auto wait_tag = [&](const int *tag) -> int {
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (__atomic_load_n(tag, __ATOMIC_ACQUIRE) != sl.seq) {
            if ((++spins & 0xfffffu) == 0) {
                if (hipStreamQuery(sl.stream) != hipErrorNotReady) {
                    DET3_TRY(h, hipStreamSynchronize(sl.stream));
                    if (__atomic_load_n(tag, __ATOMIC_ACQUIRE) == sl.seq) break;
                    h->hip_error = "the 3D detector's kernels finished without publishing their result";
                    return RDET_ERR_HIP;
                }
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) { h->hip_error = "rdet3d: no result after 10 s"; return RDET_ERR_HIP; }
            }
        }
        return RDET_OK;
    };
    int rc = wait_tag(&sl.h_out->head.seq);
    if (rc != RDET_OK) { sl.in_flight = true; return rc; }
    Det3dHead head = sl.h_out->head;
    if (head.err == D3_RETRY) {                                       // more survivors than the short front end holds: the long chain, now
        h->m_hint = head.M;
        ++h->n_retry;
        const int keep = h->path_mode;
        h->path_mode = 1;
        rc = d3_launch(h, sl);
        h->path_mode = keep;
        if (rc == RDET_OK) rc = wait_tag(&sl.h_out->head.seq);
        if (rc != RDET_OK) { sl.in_flight = true; return rc; }
        head = sl.h_out->head;
    }
    h->m_hint = head.M;
    if (head.err) { sl.in_flight = true; return head.err; }          // (whatever is left of the chain: synchronised by the next submit)
    *K = head.K;
    for (int c = 0; c < head.K; ++c) {
        rc = wait_tag(&sl.h_out->centers[c].seq);
        if (rc != RDET_OK) { sl.in_flight = true; return rc; }
        centers_xy[2 * c] = sl.h_out->centers[c].x; centers_xy[2 * c + 1] = sl.h_out->centers[c].y;
    }
    return RDET_OK;
}

int rdet3d_handle_cloud(rdet3d_t *h, double stamp, const float *xyzi, int N, float *centers_xy, int max_centers,
                        int *K, double *obs_time)
{
    if (!h || !K || N < 0 || (N > 0 && !xyzi) || max_centers < 0 || (max_centers > 0 && !centers_xy))
        return RDET_ERR_INVALID;
    *K = 0;
    if (h->n_out != 0) return RDET_ERR_INVALID;                        // (clouds submitted and not collected: collect them first)
    int rc = rdet3d_submit(h, stamp, xyzi, N, max_centers);
    if (rc != RDET_OK) { if (obs_time) *obs_time = stamp; return rc; }
    return rdet3d_collect(h, centers_xy, max_centers, K, obs_time);
}

// test hooks (include/rdet.h does not declare them): which front end the next clouds get (0: by the previous cloud's count, 1: the long chain,
// 2: the short one whatever the count), and how often each was taken
int rdet3d_debug_set_path(rdet3d_t *h, int mode)
{
    if (!h || mode < 0 || mode > 2) return RDET_ERR_INVALID;
    h->path_mode = mode;
    return RDET_OK;
}
int rdet3d_debug_path_counts(rdet3d_t *h, unsigned long long *n_short, unsigned long long *n_retry)
{
    if (!h) return RDET_ERR_INVALID;
    if (n_short) *n_short = h->n_short;
    if (n_retry) *n_retry = h->n_retry;
    return RDET_OK;
}

#ifdef RDET_DEBUG_MARKS
int rdet3d_debug_marks(rdet3d_t *h, unsigned long long *out)   // 6 x 2048 x 8
{
    for (auto &sl : h->slot) DET3_TRY(h, hipStreamSynchronize(sl.stream));
    DET3_TRY(h, hipMemcpyFromSymbol(out, HIP_SYMBOL(d3_marks_all), sizeof(unsigned long long) * 6 * 2048 * 8));
    return RDET_OK;
}
#endif

}  // extern "C"
