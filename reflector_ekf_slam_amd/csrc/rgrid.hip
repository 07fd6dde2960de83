// rgrid.hip -- MI355X-native grid-mapper front-end behind include/rgrid.h.
//
// Replaces, in the reference: sensor::VoxelFilter / AdaptiveVoxelFilter (src/sensor/voxel_filter.cc:12-120) and
// scan_matching::RealTimeCorrelativeScanMatcher2D::Match with SearchParameters / GenerateRotatedScans /
// DiscretizeScans (src/scan_matching/real_time_correlative_scan_matcher_2d.cc:20-136,
// correlative_scan_matcher_2d.cc:10-123), scored against a mapping::ProbabilityGrid given by its uint16 cells.
//
// Split of work.  Host (O(points) + O(scans), same libm as a CPU build, so every transcendental the
// discretisation depends on is bit-identical to a CPU run): the initial rotation of the cloud, the search
// parameters, one (cos, sin) pair per rotated scan.  Device (O(scans * points) + O(candidates * points)):
//   kg_discretize  rotate, translate and discretise every point of every scan into cell indices   (:86-123)
//   kg_score       one workgroup per rotated scan, one lane per translation candidate: float32 sum of the cell
//                  probabilities in point order (:20-36), the exp(-(.)^2) penalty (:127-133), workgroup arg-max with
//                  the first-maximum rule of std::max_element (:104-105)
//   kg_best        arg-max over the workgroups
// and for the voxel filters: kg_keys (voxel of every point), kg_first (a point survives iff no EARLIER point
// shares its voxel: all-pairs, the candidates arrive through the scalar cache), kg_compact (order-preserving
// ballot-scan compaction; also the range gate of the adaptive filter).  The bisection over voxel sizes of the
// adaptive filter (:47-73) stays on the host: a dozen dependent decisions on one integer each.
#include "../../include/rgrid.h"

#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {

struct BestRec { float score; int id; };

// ---- voxel filter -----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kg_keys(const float *__restrict__ xy, int n, float res, int2 *__restrict__ key,
                                               unsigned char *__restrict__ keep)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // GetCellIndex (voxel_filter.cc:105-110): RoundToInt(point / resolution), float division, lround
    key[i] = make_int2((int)lroundf(xy[2 * i] / res), (int)lroundf(xy[2 * i + 1] / res));
    keep[i] = 1;                                                       // kg_first clears it when an earlier point shares the voxel
}

// Workgroup (bi, bj), bj <= bi: the 256 points i of block bi against the 256 candidates j of block bj (j < i).
// A point that finds an earlier point in its voxel clears its flag (several workgroups may: same value).
__global__ __launch_bounds__(256) void kg_first(const int2 *__restrict__ key, int n, unsigned char *__restrict__ keep)
{
    const int bi = blockIdx.x, bj = blockIdx.y;
    if (bj > bi) return;
    const int i = bi * 256 + threadIdx.x;
    const bool live = i < n;
    const int2 k = live ? key[i] : make_int2(0, 0);
    bool dup = false;
    const int jbase = bj * 256, jend = min(n, jbase + 256);
    for (int j0 = jbase; j0 < jend; j0 += 8) {
        int2 c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = key[j0 + u];   // uniform, CONTIGUOUS addresses: one s_load_dwordx16 (a clamp per element would
                                                           // split it into eight loads); entries past n are padding, masked by j0 + u < i
#pragma unroll
        for (int u = 0; u < 8; ++u) dup |= (j0 + u < i) && c[u].x == k.x && c[u].y == k.y;
    }
    if (live && dup) keep[i] = 0;                                     // unordered_set::insert(...).second == false (:89-93)
}

// Batched variants for the adaptive filter's search: blockIdx.z / blockIdx.y picks one of up to VOX_BATCH voxel sizes,
// slice r of `key` / `keep` holds its keys / first-occurrence flags; kg_count_b sums a slice.
constexpr int VOX_BATCH = 32;
struct VoxBatch { int r; float res[VOX_BATCH]; };
__global__ __launch_bounds__(256) void kg_keys_b(const float *__restrict__ xy, int n, VoxBatch B, int2 *__restrict__ key,
                                                 unsigned char *__restrict__ keep)
{
    const int i = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (i >= n) return;
    const float res = B.res[r];
    key[(size_t)r * n + i] = make_int2((int)lroundf(xy[2 * i] / res), (int)lroundf(xy[2 * i + 1] / res));
    keep[(size_t)r * n + i] = 1;
}
__global__ __launch_bounds__(256) void kg_first_b(const int2 *__restrict__ key_all, int n, unsigned char *__restrict__ keep_all)
{
    const int bi = blockIdx.x, bj = blockIdx.y;
    if (bj > bi) return;
    const int2 *__restrict__ key = key_all + (size_t)blockIdx.z * n;
    const int i = bi * 256 + threadIdx.x;
    const bool live = i < n;
    const int2 k = live ? key[i] : make_int2(0, 0);
    bool dup = false;
    const int jbase = bj * 256, jend = min(n, jbase + 256);
    for (int j0 = jbase; j0 < jend; j0 += 8) {
        int2 c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = key[j0 + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) dup |= (j0 + u < i) && c[u].x == k.x && c[u].y == k.y;
    }
    if (live && dup) keep_all[(size_t)blockIdx.z * n + i] = 0;
}
__global__ __launch_bounds__(256) void kg_count_b(const unsigned char *__restrict__ keep_all, int n, int *__restrict__ counts)
{
    __shared__ int ws[4];
    const unsigned char *keep = keep_all + (size_t)blockIdx.x * n;
    int c = 0;
    for (int i = threadIdx.x; i < n; i += 256) c += keep[i];
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(256) void kg_range_gate(const float *__restrict__ xy, int n, float max_range,
                                                     unsigned char *__restrict__ keep)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = xy[2 * i], y = xy[2 * i + 1];
    keep[i] = (sqrtf(x * x + y * y) <= max_range) ? 1 : 0;           // FilterByMaxRange (:15-27): point.norm() <= max_range
}

// order-preserving compaction of the kept points, one workgroup, ballot scan per 1024-point tile
__global__ __launch_bounds__(1024) void kg_compact(const float *__restrict__ xy, const unsigned char *__restrict__ keep, int n,
                                                   float *__restrict__ out, int *__restrict__ count)
{
    __shared__ int wsum[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int t0 = 0; t0 < n; t0 += 1024) {
        const int i = t0 + tid;
        const bool k = i < n && keep[i];
        const unsigned long long bal = __ballot(k);
        const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int off = base, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int c = wsum[w]; if (w < wave) off += c; tot += c; }
        if (k) { const int pos = off + __popcll(bal & lt); out[2 * pos] = xy[2 * i]; out[2 * pos + 1] = xy[2 * i + 1]; }
        __syncthreads();
        if (tid == 0) base += tot;
        __syncthreads();
    }
    if (tid == 0) *count = base;
}

// ---- correlative scan matcher ---------------------------------------------------------------------
struct MatchArgs {
    int n, num_scans, num_linear, nx, ny;
    float tx, ty;                      // Eigen::Translation2f(initial translation)
    double resolution, max_x, max_y;
    double num_angular_d, step;        // orientation = (scan - num_angular) * step
    double wt, wr;
};

__global__ __launch_bounds__(256) void kg_discretize(MatchArgs A, const float *__restrict__ rot0, const float *__restrict__ cs,
                                                     int2 *__restrict__ idx)
{
#pragma clang fp contract(off)
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= A.num_scans * A.n) return;
    const int scan = e / A.n, p = e - scan * A.n;
    const float c = cs[2 * scan], s = cs[2 * scan + 1];
    const float x = rot0[2 * p], y = rot0[2 * p + 1];
    const float rx = c * x - s * y, ry = s * x + c * y;                         // Rotation2Df * point (:95-98)
    const float px = rx + A.tx, py = ry + A.ty;                                 // Affine2f(initial_translation) * point (:117-118)
    // MapLimits::GetCellIndex (map_limits.h:47-55): (x index from y, y index from x), double arithmetic, lround
    idx[e] = make_int2((int)lround((A.max_y - (double)py) / A.resolution - 0.5),
                       (int)lround((A.max_x - (double)px) / A.resolution - 0.5));
}

__device__ static inline float value_to_probability(unsigned v16)
{
#pragma clang fp contract(off)
    // probability_values.cc:11-20 (the table entry, recomputed: same two float operations), probability_values.h:53-57
    const float kMinProbability = 0.1f, kMaxProbability = 1.f - kMinProbability;
    const float lower = 1.f - kMaxProbability, upper = 1.f - kMinProbability;
    const unsigned v = v16 & 32767u;
    float cost = upper;
    if (v != 0) {
        const float kScale = (upper - lower) / (32768 - 2.f);
        cost = (float)v * kScale + (lower - kScale);
    }
    return 1.f - cost;
}

// One workgroup per rotated scan, one lane per translation candidate of that scan.  The discretised point is the
// same for the whole workgroup: its indices come through the scalar cache; the float32 sum has to run in point
// order (:27-33), so the parallelism inside a candidate is in the LOADS: the cell values of sixteen points are in
// flight before their sixteen additions.
#define KG_PF 16
__global__ __launch_bounds__(128) void kg_score(MatchArgs A, const int2 *__restrict__ idx, const unsigned short *__restrict__ cells,
                                                BestRec *__restrict__ block_best)
{
#pragma clang fp contract(off)
    __shared__ float s_sc[2];
    __shared__ int s_id[2];
    const int scan = blockIdx.x;
    const int W = 2 * A.num_linear + 1, WW = W * W;
    const int2 *__restrict__ di = idx + (size_t)scan * A.n;
    const double orientation = ((double)scan - A.num_angular_d) * A.step;
    float best = -1.f; int bid = 0x7fffffff;
    for (int r0 = 0; r0 < WW; r0 += 128) {                                         // 81 candidates per scan with the default window: one pass
        const int r = r0 + threadIdx.x;
        const bool live = r < WW;
        const int xo = (live ? r / W : 0) - A.num_linear, yo = (live ? r - (r / W) * W : 0) - A.num_linear;   // order: x offset, y offset (:64-74)
        float sum = 0.f;
        for (int p0 = 0; p0 < A.n; p0 += KG_PF) {                                  // ComputeCandidateScore (:20-36), point order
            unsigned short v[KG_PF];
            bool in[KG_PF];
#pragma unroll
            for (int u = 0; u < KG_PF; ++u) {
                const int2 c = di[p0 + u];                                         // uniform, contiguous: wide scalar loads (d_idx is padded by KG_PF)
                const int cx = c.x + xo, cy = c.y + yo;
                in[u] = cx >= 0 && cy >= 0 && cx < A.nx && cy < A.ny;
                v[u] = cells[in[u] ? A.nx * cy + cx : 0];                          // unconditional load (clamped address): a predicated
                                                                                   // one is compiled to a branch + wait per point
            }
#pragma unroll
            for (int u = 0; u < KG_PF; ++u)
                if (p0 + u < A.n) sum += in[u] ? value_to_probability(v[u]) : 0.1f;   // outside the grid: kMinProbability
        }
        sum /= (float)A.n;
        const double x = -yo * A.resolution, y = -xo * A.resolution;              // Candidate2D (correlative_scan_matcher_2d.h:62-66)
        const double a = hypot(x, y) * A.wt + fabs(orientation) * A.wr;
        const float score = (float)((double)sum * exp(-(a * a)));                  // :127-133
        const int id = scan * WW + r;
        if (live && (score > best || (score == best && id < bid))) { best = score; bid = id; }
    }
    // workgroup arg-max, first maximum (smaller id wins a tie)
    for (int off = 32; off >= 1; off >>= 1) {
        const float os = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bid, off, 64);
        if (os > best || (os == best && oi < bid)) { best = os; bid = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_sc[wave] = best; s_id[wave] = bid; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_sc[1] > best || (s_sc[1] == best && s_id[1] < bid)) { best = s_sc[1]; bid = s_id[1]; }
        block_best[blockIdx.x].score = best;
        block_best[blockIdx.x].id = bid;
    }
}

__global__ __launch_bounds__(256) void kg_best(const BestRec *__restrict__ block_best, int nblocks, BestRec *__restrict__ out)
{
    __shared__ float s_sc[256];
    __shared__ int s_id[256];
    float best = -1.f; int bid = 0x7fffffff;
    for (int b = threadIdx.x; b < nblocks; b += 256) {
        const BestRec r = block_best[b];
        if (r.score > best || (r.score == best && r.id < bid)) { best = r.score; bid = r.id; }
    }
    s_sc[threadIdx.x] = best; s_id[threadIdx.x] = bid;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const float os = s_sc[threadIdx.x + off]; const int oi = s_id[threadIdx.x + off];
            if (os > s_sc[threadIdx.x] || (os == s_sc[threadIdx.x] && oi < s_id[threadIdx.x])) { s_sc[threadIdx.x] = os; s_id[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out->score = s_sc[0]; out->id = s_id[0]; }
}

// ---- range-data inserter --------------------------------------------------------------------------
// ApplyLookupTable (probability_grid.cc:38-53): a cell without the update marker takes table[cell] (which carries the
// marker).  Concurrent lanes may race on one cell, but within a phase (hits, then misses: separate launches) every
// writer stores the SAME value table[original], and a reader sees either the original or the marked value: the plain
// 16-bit load/store pair gives the reference's result without atomics.
struct InsertArgs {
    int nx, ny, n_ret, n_miss;
    double max_x, max_y, rs;           // rs = resolution / 1000 (superscaled limits, :48-53)
    float ox, oy;
};
constexpr int SUBPX = 1000;
constexpr unsigned MARKER = 32768u;

__device__ static inline void apply_table(unsigned short *cells, int nx, int cx, int cy, const unsigned short *__restrict__ table)
{
    unsigned short *c = cells + (size_t)nx * cy + cx;
    const unsigned short v = *c;
    if (v < MARKER) *c = table[v];
}
__device__ static inline bool super_index(const InsertArgs &A, float px, float py, int &ix, int &iy)
{
    // superscaled MapLimits::GetCellIndex (map_limits.h:47-55): x index from y, y index from x
    ix = (int)lround((A.max_y - (double)py) / A.rs - 0.5);
    iy = (int)lround((A.max_x - (double)px) / A.rs - 0.5);
    return ix >= 0 && iy >= 0 && (long long)ix < (long long)A.nx * SUBPX && (long long)iy < (long long)A.ny * SUBPX;
}

// end points of all rays (returns first, then misses); *bad = 1 if anything lies outside the grid
__global__ __launch_bounds__(256) void kg_ends(InsertArgs A, const float *__restrict__ ret, const float *__restrict__ mis,
                                               int2 *__restrict__ ends, int *__restrict__ bad)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = A.n_ret + A.n_miss;
    if (i > n) return;
    int ix, iy;
    bool ok;
    if (i == n) ok = super_index(A, A.ox, A.oy, ix, iy);                        // the origin (:54-55)
    else {
        const float *p = (i < A.n_ret) ? ret + 2 * i : mis + 2 * (i - A.n_ret);
        ok = super_index(A, p[0], p[1], ix, iy);
    }
    ends[i] = make_int2(ix, iy);
    if (!ok) *bad = 1;
}
__global__ __launch_bounds__(256) void kg_hits(InsertArgs A, const int2 *__restrict__ ends, const int *__restrict__ bad,
                                               unsigned short *cells, const unsigned short *__restrict__ hit_table)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (*bad || i >= A.n_ret) return;
    apply_table(cells, A.nx, ends[i].x / SUBPX, ends[i].y / SUBPX, hit_table);   // (:59-62)
}
// One WAVE per ray.  RayToPixelMask (ray_to_pixel_mask.cc:17-168) walks the ray pixel column by pixel column and
// carries a sub-pixel ordinate `sub_y` (in units of 1 / (2 * 1000 * dx)); that recurrence has a closed form --
// with A(X) = A0 + dy * (first_pixel + 2000 * (X - X0)) the absolute ordinate at the right border of column X
// (dy * last_pixel instead of the full 2000 for the last column), column X covers the rows
//   dy > 0:  floor(A(X-1) / den) .. ceil(A(X) / den) - 1        (`while (sub_y > den)` / `if (sub_y == den)`)
//   dy <= 0: ceil(A(X-1) / den) - 1 .. floor(A(X) / den)        (`while (sub_y < 0)`  / `if (sub_y == 0)`)
// (first column: from the begin pixel's row) -- so the columns are independent.  Lanes take 64 columns at a time,
// a wave prefix sum of the rows per column flattens them into one pixel list, and the lanes share that list evenly:
// every lane has work whatever the slope.  The reference de-duplicates consecutive pixels; applying the table twice
// is a no-op, so no de-duplication here.
__global__ __launch_bounds__(256) void kg_rays(InsertArgs A, const int2 *__restrict__ ends, const int *__restrict__ bad,
                                               unsigned short *cells, const unsigned short *__restrict__ miss_table)
{
    __shared__ int s_pref[4][65], s_lo[4][64], s_step[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    const int n = A.n_ret + A.n_miss;
    if (*bad || i >= n) return;
    const long long S = SUBPX;
    long long bx = ends[n].x, by = ends[n].y, ex = ends[i].x, ey = ends[i].y;
    if (bx > ex) { long long t = bx; bx = ex; ex = t; t = by; by = ey; ey = t; }   // ordered by x (:24-27)
    const int X0 = (int)(bx / S), X1 = (int)(ex / S);
    if (X0 == X1) {                                                               // one pixel column (:35-47)
        const int y0 = (int)((by < ey ? by : ey) / S), y1 = (int)((by < ey ? ey : by) / S);
        for (int y = y0 + lane; y <= y1; y += 64) apply_table(cells, A.nx, X0, y, miss_table);
        return;
    }
    const long long dx = ex - bx, dy = ey - by, den = 2 * S * dx;
    const long long A0 = (2 * (by % S) + 1) * dx + (by / S) * den;                // absolute ordinate of the begin point (:64)
    const long long first_pixel = 2 * S - 2 * (bx % S) - 1, last_pixel = 2 * (ex % S) + 1;
    const bool up = dy > 0;
    auto a_out = [&](int X) -> long long {                                        // ordinate at the right border of column X
        return A0 + dy * (first_pixel + 2 * S * (long long)(X - X0) + ((X == X1) ? last_pixel - 2 * S : 0));
    };
    auto fdiv = [&](long long a) -> long long { return a / den; };                // a >= 0 inside the grid
    auto cdiv = [&](long long a) -> long long { return (a + den - 1) / den; };
    for (int Xc = X0; Xc <= X1; Xc += 64) {
        const int X = Xc + lane;
        int lo = 0, cnt = 0, step = 1;
        if (X <= X1) {
            const long long ao = a_out(X);
            int r_in, r_out;
            if (up) { r_in = (X == X0) ? (int)(by / S) : (int)fdiv(a_out(X - 1)); r_out = (int)cdiv(ao) - 1; cnt = r_out - r_in + 1; }
            else { r_in = (X == X0) ? (int)(by / S) : (int)cdiv(a_out(X - 1)) - 1; r_out = (int)fdiv(ao); cnt = r_in - r_out + 1; step = -1; }
            if (cnt < 1) cnt = 1;                                                 // the column's entry pixel is always visited
            lo = r_in;
        }
        // exclusive prefix sum of cnt over the wave
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
        s_pref[wave][lane + 1] = incl; s_lo[wave][lane] = lo; s_step[wave][lane] = step;
        if (lane == 0) s_pref[wave][0] = 0;
        __builtin_amdgcn_wave_barrier();
        const int total = s_pref[wave][64];
        for (int p = lane; p < total; p += 64) {
            int a = 0, b = 63;                                                    // column c with pref[c] <= p < pref[c+1]
            while (a < b) { const int mid = (a + b + 1) >> 1; if (s_pref[wave][mid] <= p) a = mid; else b = mid - 1; }
            const int row = s_lo[wave][a] + s_step[wave][a] * (p - s_pref[wave][a]);
            apply_table(cells, A.nx, Xc + a, row, miss_table);
        }
        __builtin_amdgcn_wave_barrier();
    }
}
__global__ __launch_bounds__(256) void kg_finish(unsigned short *cells, long long ncells, const int *__restrict__ bad)
{
    if (*bad) return;
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k < ncells && cells[k] >= MARKER) cells[k] -= MARKER;                    // Grid2D::FinishUpdate (grid_2d.cc:20-29)
}

// Grid2D::GrowLimits' cell copy (grid_2d.cc:81-91): every cell of the grown grid in one pass -- the old value
// inside the window at (off_x, off_y), unknown (0) outside
__global__ __launch_bounds__(256) void kg_grow(const unsigned short *__restrict__ old_cells, int nx, int ny, unsigned short *__restrict__ new_cells,
                                               int nnx, int nny, int off_x, int off_y)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= nnx) return;
    const int ox = x - off_x, oy = y - off_y;
    const bool in = ox >= 0 && oy >= 0 && ox < nx && oy < ny;
    new_cells[(size_t)y * nnx + x] = in ? old_cells[(size_t)oy * nx + ox] : (unsigned short)0;
}

// ---------------------------------------------------------------------------------------------------------------
// CeresScanMatcher2D::Match (src/scan_matching/ceres_scan_matcher_2d.cc:26-62) -- the whole Levenberg-Marquardt solve in
// ONE workgroup: a 3-parameter problem whose only wide part is the sum over the points (a few hundred after the
// adaptive voxel filter).  Every iteration the threads evaluate their points at the candidate pose (bicubic
// interpolation of the correspondence cost, 16 cell reads, analytic chain rule instead of Ceres' jets) and reduce
// 1/2|r|^2, J'r and J'J (10 doubles) through shuffles + LDS; thread 0 then runs the scalar trust-region logic on a
// state that lives in LDS (so the kernel fits 1024 threads: one point per thread up to 1024 points) and publishes
// the next candidate -- no host round trip, no second launch.
// Ceres is not in the image and not pinned by the reference: the algorithm is restated from its published sources
// (DESIGN.md 4b lists them); parity with a Ceres build is unpinned.
struct RefineArgs {
    int nx, ny, n, max_iter, max_nonmono;
    double res, max_x, max_y, w_occ, w_t, w_r, tx, ty, x0, y0, a0;
};
struct RefineOut {
    double pose[3]; double initial_cost, final_cost; int iterations, termination;
#ifdef RGRID_DEBUG_TIMING
    long long dbg[16];                                 // cycles per phase, summed over the iterations (thread 0)
#endif
};
#ifdef RGRID_DEBUG_TIMING
__device__ static inline long long pinned_clock() { __builtin_amdgcn_sched_barrier(0); const long long t = clock64(); __builtin_amdgcn_sched_barrier(0); return t; }
#define RT_MARK(k) do { if (threadIdx.x == 0) { const long long t_ = pinned_clock(); dbg[k] += t_ - tprev; tprev = t_; } } while (0)
#else
#define RT_MARK(k) do { } while (0)
#endif
struct RefineState {
    double x[3], xc[3], g[3], H[6], s[3], best[3];
    double x_cost, x_norm, gmax, radius, decrease, mcc, min_cost, initial_cost;
    double ev_min, ev_cur, ev_ref, ev_cand, acc_ref, acc_cand;              // TrustRegionStepEvaluator
    int nonmono, invalid, iter, termination, successful, done;
};
constexpr double REFINE_PAD = 536870911.0;                                      // kPadding = INT_MAX / 4 (occupied_space_cost_function_2d.cc:57)

__device__ static inline float value_to_cost(unsigned v16)
{
#pragma clang fp contract(off)
    const float kMinProbability = 0.1f, kMaxProbability = 1.f - kMinProbability;
    const float lower = 1.f - kMaxProbability, upper = 1.f - kMinProbability;
    const unsigned v = v16 & 32767u;
    const float kScale = (upper - lower) / (32768 - 2.f);
    const float c = (float)v * kScale + (lower - kScale);
    return v == 0 ? upper : c;
}
__device__ static inline void hermite(double p0, double p1, double p2, double p3, double x, double &f, double &dfdx)
{
#pragma clang fp contract(off)
    const double a = 0.5 * (-p0 + 3.0 * p1 - 3.0 * p2 + p3);                    // ceres::CubicHermiteSpline
    const double b = 0.5 * (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3);
    const double c = 0.5 * (-p0 + p2);
    f = p1 + x * (c + x * (b + x * a));
    dfdx = c + x * (2.0 * b + 3.0 * a * x);
}
// All-lanes sum of a double over the wave without touching LDS: DPP inside rows of 16 (quad permutes, half-row and row
// mirror), then gfx950's v_permlane16_swap / v_permlane32_swap across the rows (each returns both halves of the exchange,
// so one instruction pair per step serves the low and the high dword).  Six steps, no ds_bpermute, no waitcnt.
template <int CTRL>
__device__ static inline double mov_dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ static inline double wave_sum_f64(double v)
{
    v += mov_dpp_f64<0xB1>(v);                          // quad_perm [1,0,3,2]: lane ^ 1
    v += mov_dpp_f64<0x4E>(v);                          // quad_perm [2,3,0,1]: lane ^ 2
    v += mov_dpp_f64<0x141>(v);                         // row_half_mirror: the other quad of the half row
    v += mov_dpp_f64<0x140>(v);                         // row_mirror: the other half row
    {
        auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
        auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
        v = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);      // rows 0+1, 2+3
    }
    {
        auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
        auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
        v = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);      // both halves of the wave
    }
    return v;
}

// Partial sums of this wave into part[wave][]: [0] = |r|^2, [1..3] = J'r, [4..9] = J'J (xx xy xt yy yt tt)
#ifdef RGRID_DEBUG_TIMING
__shared__ long long edbg[8];
#define ET_MARK(k) do { if (threadIdx.x == 0) { const long long t_ = pinned_clock(); edbg[k] += t_ - et; et = t_; } } while (0)
#else
#define ET_MARK(k) do { } while (0)
#endif
__device__ static void refine_eval(const RefineArgs &A, const unsigned short *__restrict__ cells, const float *__restrict__ pts,
                                   const double p0, const double p1, const double p2, double (*part)[10])
{
#pragma clang fp contract(off)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef RGRID_DEBUG_TIMING
    long long et = pinned_clock();
#endif
    double sn, cs;
    sincos(p2, &sn, &cs);
    ET_MARK(0);
    const double scale = A.w_occ / sqrt((double)A.n);
    const double ninv = -1.0 / A.res;                                            // d(row) / d(world x): Jet / scalar = * (1 / scalar)
    double acc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = 0.;
    for (int i = tid; i < A.n; i += blockDim.x) {
        const float2 pt = reinterpret_cast<const float2 *>(pts)[i];
        const double px = (double)pt.x, py = (double)pt.y;
        const double wx = cs * px - sn * py + p0, wy = sn * px + cs * py + p1;
        const double dwx = -sn * px - cs * py, dwy = cs * px - sn * py;          // d world / d angle
        const double r = (A.max_x - wx) / A.res - 0.5 + REFINE_PAD, q = (A.max_y - wy) / A.res - 0.5 + REFINE_PAD;
        const double rf = floor(r), qf = floor(q);
        // cell (x = col, y = row) of the interpolation's base corner, as int32 (clamped far outside the grid first)
        const int row = (int)fmin(fmax(rf - REFINE_PAD, -8.), (double)A.ny + 8.), col = (int)fmin(fmax(qf - REFINE_PAD, -8.), (double)A.nx + 8.);
        double f[4], dq[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            double v[4];
            const int y = row - 1 + a;
            const int yc = min(max(y, 0), A.ny - 1);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int x = col - 1 + b;
                const bool in = x >= 0 && y >= 0 && x < A.nx && y < A.ny;        // GridArrayAdapter::GetValue (:69-81)
                const float cv = value_to_cost(cells[A.nx * yc + min(max(x, 0), A.nx - 1)]);   // unconditional clamped load, select after
                v[b] = (double)(in ? cv : 0.9f);
            }
            hermite(v[0], v[1], v[2], v[3], q - qf, f[a], dq[a]);
        }
        ET_MARK(1);
        double val, dvdr, dvdq, unused;
        hermite(f[0], f[1], f[2], f[3], r - rf, val, dvdr);
        hermite(dq[0], dq[1], dq[2], dq[3], r - rf, dvdq, unused);
        const double ri = scale * val;
        const double J0 = scale * (dvdr * ninv), J1 = scale * (dvdq * ninv);
        const double J2 = scale * (dvdr * (dwx * ninv) + dvdq * (dwy * ninv));
        acc[0] += ri * ri;
        acc[1] += J0 * ri; acc[2] += J1 * ri; acc[3] += J2 * ri;
        acc[4] += J0 * J0; acc[5] += J0 * J1; acc[6] += J0 * J2; acc[7] += J1 * J1; acc[8] += J1 * J2; acc[9] += J2 * J2;
        ET_MARK(2);
    }
    ET_MARK(3);
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = wave_sum_f64(acc[k]);
    if (lane == 0)
        for (int k = 0; k < 10; ++k) part[wave][k] = acc[k];
    ET_MARK(4);
}
__device__ static inline bool chol3(const double A[6], const double b[3], double y[3])
{
#pragma clang fp contract(off)
    const double l00 = sqrt(A[0]);
    if (!(l00 > 0.)) return false;
    const double l10 = A[1] / l00, l20 = A[2] / l00;
    const double d1 = A[3] - l10 * l10;
    if (!(d1 > 0.)) return false;
    const double l11 = sqrt(d1), l21 = (A[4] - l20 * l10) / l11;
    const double d2 = A[5] - l20 * l20 - l21 * l21;
    if (!(d2 > 0.)) return false;
    const double l22 = sqrt(d2);
    const double z0 = b[0] / l00, z1 = (b[1] - l10 * z0) / l11, z2 = (b[2] - l20 * z0 - l21 * z1) / l22;
    y[2] = z2 / l22; y[1] = (z1 - l21 * y[2]) / l11; y[0] = (z0 - l10 * y[1] - l20 * y[2]) / l00;
    return isfinite(y[0]) && isfinite(y[1]) && isfinite(y[2]);
}
// First wave: totals of an evaluation at pose p (the wave partials + the translation / rotation delta blocks,
// translation_delta_cost_functor_2d.h:24-29, rotation_delta_cost_functor_2d.h:24-28)
__device__ static void refine_totals(const RefineArgs &A, const double (*part)[10], int nw, const double p[3], double S[10])
{
#pragma clang fp contract(off)
    // called by the whole first wave: lane k < 10 adds sum k over the waves (in wave order), lane 0 collects them
    {
        const int lane = threadIdx.x & 63, k = lane < 10 ? lane : 0;
        double v = 0.;
        for (int w = 0; w < nw; ++w) v += part[w][k];
#pragma unroll
        for (int q = 0; q < 10; ++q) S[q] = __shfl(v, q, 64);
    }
    const double r0 = A.w_t * (p[0] - A.tx), r1 = A.w_t * (p[1] - A.ty), r2 = A.w_r * (p[2] - A.a0);
    S[0] += r0 * r0 + r1 * r1 + r2 * r2;
    S[1] += A.w_t * r0; S[2] += A.w_t * r1; S[3] += A.w_r * r2;
    S[4] += A.w_t * A.w_t; S[7] += A.w_t * A.w_t; S[9] += A.w_r * A.w_r;
}
// Thread 0: TrustRegionMinimizer's loop head up to the next candidate -- FinalizeIterationAndCheckIfMinimizerCanContinue,
// LevenbergMarquardtStrategy::ComputeStep on the column-scaled Jacobian (through the normal equations), the model cost
// change; invalid steps shrink the radius and retry without a new evaluation.  Sets st.xc / st.mcc or st.done.
__device__ static void refine_next_candidate(const RefineArgs &A, RefineState &st)
{
#pragma clang fp contract(off)
    for (;;) {
        if (st.successful && st.x_cost < st.min_cost) { st.min_cost = st.x_cost; st.best[0] = st.x[0]; st.best[1] = st.x[1]; st.best[2] = st.x[2]; }
        if (st.iter >= A.max_iter) { st.termination = 1; st.done = 1; return; }
        if (st.successful && st.gmax <= 1e-10) { st.termination = 0; st.done = 1; return; }
        if (st.radius < 1e-32) { st.termination = 0; st.done = 1; return; }
        ++st.iter;
        const double s0 = st.s[0], s1 = st.s[1], s2 = st.s[2];
        const double Hs[6] = {s0 * st.H[0] * s0, s0 * st.H[1] * s1, s0 * st.H[2] * s2, s1 * st.H[3] * s1, s1 * st.H[4] * s2, s2 * st.H[5] * s2};
        const double gs[3] = {s0 * st.g[0], s1 * st.g[1], s2 * st.g[2]};
        double M[6] = {Hs[0], Hs[1], Hs[2], Hs[3], Hs[4], Hs[5]}, y[3], step[3] = {0., 0., 0.};
        M[0] += fmin(fmax(Hs[0], 1e-6), 1e32) / st.radius; M[3] += fmin(fmax(Hs[3], 1e-6), 1e32) / st.radius; M[5] += fmin(fmax(Hs[5], 1e-6), 1e32) / st.radius;
        double mcc = -1.;
        if (chol3(M, gs, y)) {
            step[0] = -y[0]; step[1] = -y[1]; step[2] = -y[2];
            const double Hd[3] = {Hs[0] * step[0] + Hs[1] * step[1] + Hs[2] * step[2], Hs[1] * step[0] + Hs[3] * step[1] + Hs[4] * step[2],
                                  Hs[2] * step[0] + Hs[4] * step[1] + Hs[5] * step[2]};
            mcc = -(step[0] * gs[0] + step[1] * gs[1] + step[2] * gs[2]) - 0.5 * (step[0] * Hd[0] + step[1] * Hd[1] + step[2] * Hd[2]);
        }
        if (!(mcc > 0.)) {                                                       // HandleInvalidStep
            st.successful = 0;
            if (++st.invalid >= 5) { st.termination = 2; st.done = 1; return; }
            st.radius /= st.decrease; st.decrease *= 2.;
            continue;
        }
        st.invalid = 0;
        st.mcc = mcc;
        st.xc[0] = st.x[0] + step[0] * s0; st.xc[1] = st.x[1] + step[1] * s1; st.xc[2] = st.x[2] + step[2] * s2;
        return;
    }
}
// Thread 0: the candidate's evaluation is in -- tolerances, step quality, accept / reject (HandleSuccessfulStep /
// HandleUnsuccessfulStep, LevenbergMarquardtStrategy::StepAccepted / StepRejected, TrustRegionStepEvaluator)
__device__ static void refine_judge(const RefineArgs &A, RefineState &st, const double S[10])
{
#pragma clang fp contract(off)
    const double c_cost = 0.5 * S[0];
    const double d0 = st.x[0] - st.xc[0], d1 = st.x[1] - st.xc[1], d2 = st.x[2] - st.xc[2];
    if (sqrt(d0 * d0 + d1 * d1 + d2 * d2) <= 1e-8 * (st.x_norm + 1e-8)) { st.termination = 0; st.done = 1; return; }   // ParameterToleranceReached
    if (fabs(st.x_cost - c_cost) <= 1e-6 * st.x_cost) { st.termination = 0; st.done = 1; return; }                      // FunctionToleranceReached
    const double rho = fmax((st.ev_cur - c_cost) / st.mcc, (st.ev_ref - c_cost) / (st.acc_ref + st.mcc));               // StepQuality
    if (rho > 1e-3) {
        st.x[0] = st.xc[0]; st.x[1] = st.xc[1]; st.x[2] = st.xc[2];
        st.g[0] = S[1]; st.g[1] = S[2]; st.g[2] = S[3];
        for (int k = 0; k < 6; ++k) st.H[k] = S[4 + k];
        st.x_norm = sqrt(st.x[0] * st.x[0] + st.x[1] * st.x[1] + st.x[2] * st.x[2]);
        st.x_cost = c_cost;
        st.gmax = fmax(fabs(S[1]), fmax(fabs(S[2]), fabs(S[3])));
        st.successful = 1;
        const double t = 2. * rho - 1.;
        st.radius = fmin(1e16, st.radius / fmax(1. / 3., 1. - t * t * t));
        st.decrease = 2.;
        st.ev_cur = c_cost; st.acc_cand += st.mcc; st.acc_ref += st.mcc;
        if (st.ev_cur < st.ev_min) { st.ev_min = st.ev_cur; st.nonmono = 0; st.ev_cand = st.ev_cur; st.acc_cand = 0.; }
        else { ++st.nonmono; if (st.ev_cur > st.ev_cand) { st.ev_cand = st.ev_cur; st.acc_cand = 0.; } }
        if (st.nonmono == A.max_nonmono) { st.ev_ref = st.ev_cand; st.acc_ref = st.acc_cand; }
    } else {
        st.successful = 0;
        st.radius /= st.decrease; st.decrease *= 2.;
    }
}
__global__ __launch_bounds__(1024) void kg_refine(RefineArgs A, const unsigned short *__restrict__ cells, const float *__restrict__ pts,
                                                  RefineOut *__restrict__ out)
{
#pragma clang fp contract(off)
    __shared__ double part[16][10];
    __shared__ RefineState st;
    const int nw = blockDim.x >> 6;
    refine_eval(A, cells, pts, A.x0, A.y0, A.a0, part);                          // IterationZero
    __syncthreads();
    double S[10];
    if (threadIdx.x < 64) {
        const double x[3] = {A.x0, A.y0, A.a0};
        refine_totals(A, part, nw, x, S);
    }
    if (threadIdx.x == 0) {
        const double x[3] = {A.x0, A.y0, A.a0};
        for (int k = 0; k < 3; ++k) { st.x[k] = st.xc[k] = st.best[k] = x[k]; st.g[k] = S[1 + k]; }
        for (int k = 0; k < 6; ++k) st.H[k] = S[4 + k];
        st.s[0] = 1. / (1. + sqrt(S[4])); st.s[1] = 1. / (1. + sqrt(S[7])); st.s[2] = 1. / (1. + sqrt(S[9]));   // Jacobi scaling, fixed
        st.x_cost = st.initial_cost = 0.5 * S[0];
        st.x_norm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        st.gmax = fmax(fabs(S[1]), fmax(fabs(S[2]), fabs(S[3])));
        st.radius = 1e4; st.decrease = 2.; st.mcc = 0.; st.min_cost = INFINITY;
        st.ev_min = st.ev_cur = st.ev_ref = st.ev_cand = st.x_cost; st.acc_ref = st.acc_cand = 0.;
        st.nonmono = st.invalid = st.iter = 0; st.termination = 1; st.successful = 1; st.done = 0;
        refine_next_candidate(A, st);
    }
    __syncthreads();
#ifdef RGRID_DEBUG_TIMING
    if (threadIdx.x < 8) edbg[threadIdx.x] = 0;
    __syncthreads();
    long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = pinned_clock();
#endif
    while (!st.done) {
        const double c0 = st.xc[0], c1 = st.xc[1], c2 = st.xc[2];
        RT_MARK(0);                                                              // loop head: LDS read of the candidate
        refine_eval(A, cells, pts, c0, c1, c2, part);
        RT_MARK(1);                                                              // evaluation + wave reduction
        __syncthreads();
        RT_MARK(2);                                                              // barrier
        if (threadIdx.x < 64) {
            const double xc[3] = {c0, c1, c2};
            refine_totals(A, part, nw, xc, S);
        }
        if (threadIdx.x == 0) {
            RT_MARK(3);
            refine_judge(A, st, S);
            RT_MARK(4);
            if (!st.done) refine_next_candidate(A, st);
            RT_MARK(5);
        }
        __syncthreads();
        RT_MARK(6);
    }
    if (threadIdx.x == 0) {
        out->pose[0] = st.best[0]; out->pose[1] = st.best[1]; out->pose[2] = st.best[2];
        out->initial_cost = st.initial_cost; out->final_cost = st.min_cost; out->iterations = st.iter; out->termination = st.termination;
#ifdef RGRID_DEBUG_TIMING
        for (int k = 0; k < 8; ++k) { out->dbg[k] = dbg[k]; out->dbg[8 + k] = edbg[k]; }
#endif
    }
}

// ProbabilityGrid::DrawToSubmapTexture (probability_grid.cc:86-131): bounding box of the known cells
// (Grid2D::ComputeCroppedLimits -- the box ApplyLookupTable keeps is the box of the cells that are not 0) ...
__global__ __launch_bounds__(256) void kg_known_box(const unsigned short *__restrict__ cells, int nx, int ny, int *__restrict__ box)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    const bool known = x < nx && cells[(size_t)nx * y + x] != 0;
    const unsigned long long m = __ballot(known);
    if (m == 0) return;
    if ((threadIdx.x & 63) == 0) {                                               // one lane per wave: its lanes' x range, the row
        const int base = blockIdx.x * 256 + (threadIdx.x & ~63);
        atomicMin(&box[0], base + __ffsll((long long)m) - 1);
        atomicMax(&box[2], base + 63 - __clzll((long long)m));
        atomicMin(&box[1], y);
        atomicMax(&box[3], y);
    }
}
// ... and two bytes (value, alpha) per cell of that window, x fastest, through a 32768-entry table of byte pairs
__global__ __launch_bounds__(256) void kg_texture(const unsigned short *__restrict__ cells, int nx, int x0, int y0, int w,
                                                  const unsigned short *__restrict__ table, unsigned short *__restrict__ out)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x < w) out[(size_t)w * y + x] = table[cells[(size_t)nx * (y0 + y) + (x0 + x)] & 32767u];
}

// (value, alpha) of every cell value: 128 - ProbabilityToLogOddsInteger(GetProbability) (probability_grid.cc:97-114,
// submaps.h:22-41), host float32 with the same libm a CPU build uses; entry = value | alpha << 8
void texture_table(unsigned short *table)
{
#pragma clang fp contract(off)
    const float kMinP = 0.1f, kMaxP = 1.f - kMinP;
    const float kMaxLogOdds = std::log(kMaxP / (1.f - kMaxP)), kMinLogOdds = std::log(kMinP / (1.f - kMinP));
    const float lower = 1.f - kMaxP, upper = 1.f - kMinP, kScale = (upper - lower) / (32768 - 2.f);
    table[0] = 0;                                                                // unknown: (0, 0)
    for (int v = 1; v < 32768; ++v) {
        const float cost = (float)v * kScale + (lower - kScale);
        const float p = 1.f - cost;
        const float logit = std::log(p / (1.f - p));
        const int li = (int)std::lround((logit - kMinLogOdds) * 254.f / (kMaxLogOdds - kMinLogOdds)) + 1;
        const int delta = 128 - li;
        const unsigned alpha = (unsigned)(delta > 0 ? 0 : -delta) & 255u, value = (unsigned)(delta > 0 ? delta : 0) & 255u;
        table[v] = (unsigned short)(value | ((value || alpha) ? alpha : 1u) << 8);
    }
}

// ComputeLookupTableToApplyCorrespondenceCostOdds(Odds(probability)) (probability_values.cc:76-96), host float32
void lookup_table(float probability, unsigned short *table)
{
#pragma clang fp contract(off)
    const float kMinProbability = 0.1f, kMaxProbability = 1.f - kMinProbability;
    const float lower = 1.f - kMaxProbability, upper = 1.f - kMinProbability;
    auto cost_to_value = [&](float c) -> unsigned short {                        // BoundedFloatToValue (probability_values.h:15-29)
        float cl = c;
        if (cl > upper) cl = upper;
        if (cl < lower) cl = lower;
        return (unsigned short)((int)std::lround((cl - lower) * (32766.f / (upper - lower))) + 1);
    };
    const float odds = probability / (1.f - probability);
    {
        const float p = odds / (odds + 1.f);
        table[0] = (unsigned short)(cost_to_value(1.f - p) + MARKER);
    }
    const float kScale = (upper - lower) / (32768 - 2.f);
    for (int cell = 1; cell != 32768; ++cell) {
        const float cost = cell * kScale + (lower - kScale);                     // kValueToCorrespondenceCost[cell]
        const float pc = 1.f - cost;
        const float o = odds * (pc / (1.f - pc));
        const float p = o / (o + 1.f);
        table[cell] = (unsigned short)(cost_to_value(1.f - p) + MARKER);
    }
}

// Project2D(Rigid3f::Rotation(AngleAxisf(angle, UnitZ))) as a (cos, sin) pair, restating Eigen 3.3 in float32:
// Quaternionf(AngleAxisf) = (cos(a/2), 0, 0, sin(a/2)); GetYaw (transform.h:27-33) = atan2 of q * UnitX with
// Eigen's  v + w*uv + vec x uv,  uv = 2 (vec x v); Rotation2Df(yaw) rotates with (cos yaw, sin yaw).  Host libm.
void rotation_cs(float angle, float *c, float *s)
{
#pragma clang fp contract(off)
    const float ha = 0.5f * angle;
    const float w = std::cos(ha), z = std::sin(ha);
    const float uvy = z + z;
    const float dx = (1.f + w * 0.f) + (0.f * 0.f - z * uvy);
    const float dy = (0.f + w * uvy) + (z * 0.f - 0.f * 0.f);
    const float yaw = std::atan2(dy, dx);
    *c = std::cos(yaw); *s = std::sin(yaw);
}

}  // namespace

struct rgrid {
    int max_points, max_cells, max_candidates, device;
    hipStream_t stream;
    float *d_in, *d_a, *d_b, *d_cs;       // points in / two result buffers / per-scan (cos, sin)
    int2 *d_key, *d_idx;
    int2 *d_key_b;                              // VOX_BATCH key slices (adaptive filter search)
    unsigned char *d_keep_b;                    // VOX_BATCH + 1 flag slices (the last one parks the current result)
    int *d_counts, *h_counts;
    unsigned char *d_keep;
    unsigned short *d_cells, *d_hit, *d_miss;   // grid; hit / miss lookup tables (uint16[32768])
    unsigned short *d_cells2;                   // second grid buffer: target of a growth (then swapped with d_cells), texture staging
    unsigned short *d_tex;                      // (value, alpha) table of the texture export, built on first use
    int *d_box, *h_box;
    float *d_mis;                               // misses of an insertion
    int2 *d_ends;
    int *d_bad;
    float tab_hit_p, tab_miss_p;                // probabilities the resident tables were built for
    BestRec *d_bb, *d_best;
    int *d_count;
    // pinned staging
    float *h_pts;
    int *h_count;
    BestRec *h_best;
    RefineOut *d_refine, *h_refine;
    // grid
    int nx, ny;
    double resolution, max_x, max_y;
    bool have_grid;
    std::string hip_error;
};

#define G_TRY(h, expr)                                                              \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            if (h) (h)->hip_error = std::string(#expr) + ": " + hipGetErrorString(e_); \
            return RGRID_ERR_HIP;                                                   \
        }                                                                           \
    } while (0)

namespace {

// VoxelFilter(res).Filter on the device buffer `src` (n points) into `dst`; *m = survivors (synchronises)
int voxel_pass(rgrid_t *h, const float *src, int n, float res, float *dst, int *m)
{
    if (n == 0) { *m = 0; return RGRID_OK; }
    const int blocks = (n + 255) / 256;
    hipLaunchKernelGGL(kg_keys, dim3(blocks), dim3(256), 0, h->stream, src, n, res, h->d_key, h->d_keep);
    hipLaunchKernelGGL(kg_first, dim3(blocks, blocks), dim3(256), 0, h->stream, h->d_key, n, h->d_keep);
    hipLaunchKernelGGL(kg_compact, dim3(1), dim3(1024), 0, h->stream, src, h->d_keep, n, dst, h->d_count);
    G_TRY(h, hipMemcpyAsync(h->h_count, h->d_count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    G_TRY(h, hipStreamSynchronize(h->stream));
    *m = *h->h_count;
    return RGRID_OK;
}

// Survivor counts of VoxelFilter(res[r]) for r < R <= VOX_BATCH on the device cloud `src`, one round trip; the
// first-occurrence flags stay in slice r of d_keep_b for voxel_emit
int voxel_counts(rgrid_t *h, const float *src, int n, const float *res, int R, int *counts)
{
    VoxBatch B;
    B.r = R;
    for (int r = 0; r < R; ++r) B.res[r] = res[r];
    const int blocks = (n + 255) / 256;
    hipLaunchKernelGGL(kg_keys_b, dim3(blocks, R), dim3(256), 0, h->stream, src, n, B, h->d_key_b, h->d_keep_b);
    hipLaunchKernelGGL(kg_first_b, dim3(blocks, blocks, R), dim3(256), 0, h->stream, h->d_key_b, n, h->d_keep_b);
    hipLaunchKernelGGL(kg_count_b, dim3(R), dim3(256), 0, h->stream, h->d_keep_b, n, h->d_counts);
    G_TRY(h, hipMemcpyAsync(h->h_counts, h->d_counts, sizeof(int) * R, hipMemcpyDeviceToHost, h->stream));
    G_TRY(h, hipStreamSynchronize(h->stream));
    for (int r = 0; r < R; ++r) counts[r] = h->h_counts[r];
    return RGRID_OK;
}

int upload_points(rgrid_t *h, const float *xy, int n)
{
    std::memcpy(h->h_pts, xy, sizeof(float) * 2 * (size_t)n);
    G_TRY(h, hipMemcpyAsync(h->d_in, h->h_pts, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, h->stream));
    return RGRID_OK;
}

int download_points(rgrid_t *h, const float *d_src, int m, float *out_xy, int out_cap)
{
    if (m > out_cap) return RGRID_ERR_BUFFER;
    if (m > 0) {
        G_TRY(h, hipMemcpyAsync(h->h_pts, d_src, sizeof(float) * 2 * (size_t)m, hipMemcpyDeviceToHost, h->stream));
        G_TRY(h, hipStreamSynchronize(h->stream));
        std::memcpy(out_xy, h->h_pts, sizeof(float) * 2 * (size_t)m);
    }
    return RGRID_OK;
}

}  // namespace

extern "C" {

int rgrid_abi_version(void) { return RGRID_ABI_VERSION; }

int rgrid_draw_texture(rgrid_t *h, uint8_t *cells, long cap, int box[4], double slice_max[2])
{
    if (!h || !cells || !box || !slice_max || !h->have_grid) return RGRID_ERR_INVALID;
    G_TRY(h, hipSetDevice(h->device));
    if (!h->d_tex) {
        std::vector<unsigned short> t(32768);
        texture_table(t.data());
        G_TRY(h, hipMalloc(&h->d_tex, 2 * 32768));
        G_TRY(h, hipMemcpy(h->d_tex, t.data(), 2 * 32768, hipMemcpyHostToDevice));
    }
    h->h_box[0] = h->h_box[1] = INT_MAX; h->h_box[2] = h->h_box[3] = -1;
    G_TRY(h, hipMemcpyAsync(h->d_box, h->h_box, 4 * sizeof(int), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(kg_known_box, dim3((h->nx + 255) / 256, h->ny), dim3(256), 0, h->stream, h->d_cells, h->nx, h->ny, h->d_box);
    G_TRY(h, hipMemcpyAsync(h->h_box, h->d_box, 4 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    G_TRY(h, hipStreamSynchronize(h->stream));
    int x0 = h->h_box[0], y0 = h->h_box[1], x1 = h->h_box[2], y1 = h->h_box[3];
    if (x1 < 0) { x0 = y0 = 0; x1 = y1 = 0; }                                     // nothing known: offset 0, CellLimits(1, 1) (grid_2d.cc:39-44)
    const int w = x1 - x0 + 1, hh = y1 - y0 + 1;
    box[0] = x0; box[1] = y0; box[2] = w; box[3] = hh;
    slice_max[0] = h->max_x - h->resolution * y0;                                // (:122-123)
    slice_max[1] = h->max_y - h->resolution * x0;
    if (2l * w * hh > cap) return RGRID_ERR_BUFFER;
    hipLaunchKernelGGL(kg_texture, dim3((w + 255) / 256, hh), dim3(256), 0, h->stream, h->d_cells, h->nx, x0, y0, w, h->d_tex, h->d_cells2);
    G_TRY(h, hipGetLastError());
    G_TRY(h, hipMemcpyAsync(cells, h->d_cells2, 2 * (size_t)w * hh, hipMemcpyDeviceToHost, h->stream));
    G_TRY(h, hipStreamSynchronize(h->stream));
    return RGRID_OK;
}

int rgrid_refine_match(rgrid_t *h, const rgrid_refine_options *opt, const double target_translation[2], const double initial_pose[3],
                       const float *points_xy, int n, double pose_estimate[3], rgrid_refine_summary *summary)
{
    if (!h || !opt || !target_translation || !initial_pose || !pose_estimate || n < 0 || (n > 0 && !points_xy)) return RGRID_ERR_INVALID;
    if (!h->have_grid || !(opt->occupied_space_weight > 0.) || !(opt->translation_weight > 0.) || !(opt->rotation_weight > 0.) ||
        opt->max_num_iterations < 0)
        return RGRID_ERR_INVALID;                                                // the reference CHECK_GTs the weights (ceres_scan_matcher_2d.cc:37,47,52)
    if (n == 0) return RGRID_ERR_EMPTY;
    if (n > h->max_points) return RGRID_ERR_CAPACITY;
    G_TRY(h, hipSetDevice(h->device));
    std::memcpy(h->h_pts, points_xy, sizeof(float) * 2 * (size_t)n);
    G_TRY(h, hipMemcpyAsync(h->d_in, h->h_pts, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, h->stream));
    RefineArgs A;
    A.nx = h->nx; A.ny = h->ny; A.n = n; A.max_iter = opt->max_num_iterations; A.max_nonmono = opt->use_nonmonotonic_steps ? 5 : 0;
    A.res = h->resolution; A.max_x = h->max_x; A.max_y = h->max_y;
    A.w_occ = opt->occupied_space_weight; A.w_t = opt->translation_weight; A.w_r = opt->rotation_weight;
    A.tx = target_translation[0]; A.ty = target_translation[1];
    A.x0 = initial_pose[0]; A.y0 = initial_pose[1]; A.a0 = initial_pose[2];
    const int threads = std::min(1024, ((n + 63) / 64) * 64);                      // no idle waves in the barriers and the partial sums
    hipLaunchKernelGGL(kg_refine, dim3(1), dim3(threads), 0, h->stream, A, h->d_cells, h->d_in, h->d_refine);
    G_TRY(h, hipGetLastError());
    G_TRY(h, hipMemcpyAsync(h->h_refine, h->d_refine, sizeof(RefineOut), hipMemcpyDeviceToHost, h->stream));
    G_TRY(h, hipStreamSynchronize(h->stream));
    pose_estimate[0] = h->h_refine->pose[0]; pose_estimate[1] = h->h_refine->pose[1]; pose_estimate[2] = h->h_refine->pose[2];
    if (summary) {
        summary->initial_cost = h->h_refine->initial_cost; summary->final_cost = h->h_refine->final_cost;
        summary->iterations = h->h_refine->iterations; summary->termination = h->h_refine->termination;
    }
    return RGRID_OK;
}

#ifdef RGRID_DEBUG_TIMING
int rgrid_debug_refine_cycles(rgrid_t *h, long long out[16])
{
    for (int k = 0; k < 16; ++k) out[k] = h->h_refine->dbg[k];
    return RGRID_OK;
}
#endif

const char *rgrid_strerror(int code)
{
    switch (code) {
    case RGRID_OK: return "ok";
    case RGRID_ERR_INVALID: return "invalid argument";
    case RGRID_ERR_HIP: return "HIP runtime error";
    case RGRID_ERR_CAPACITY: return "capacity of the handle exceeded";
    case RGRID_ERR_BUFFER: return "caller buffer too small";
    case RGRID_ERR_EMPTY: return "empty point cloud";
    default: return "unknown error";
    }
}

const char *rgrid_last_hip_error(rgrid_t *h) { return h ? h->hip_error.c_str() : ""; }

int rgrid_create(int max_points, int max_cells, int max_candidates, int device, rgrid_t **out)
{
    if (!out || max_points < 1 || max_cells < 1 || max_candidates < 1) return RGRID_ERR_INVALID;
    *out = nullptr;
    rgrid_t *h = new (std::nothrow) rgrid();
    if (!h) return RGRID_ERR_INVALID;
    h->max_points = max_points; h->max_cells = max_cells; h->max_candidates = max_candidates; h->device = device;
    h->have_grid = false;
    h->tab_hit_p = h->tab_miss_p = -1.f;
    const size_t np = (size_t)max_points;
    // discretised scans: every candidate's scan must fit: num_scans <= max_candidates, num_scans * n <= np * scans_cap
    int rc = [&]() -> int {
        G_TRY(h, hipSetDevice(device));
        G_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        G_TRY(h, hipMalloc(&h->d_in, 8 * np)); G_TRY(h, hipMalloc(&h->d_a, 8 * np)); G_TRY(h, hipMalloc(&h->d_b, 8 * np));
        G_TRY(h, hipMalloc(&h->d_key, 8 * np + 64)); G_TRY(h, hipMemset(h->d_key, 0, 8 * np + 64)); G_TRY(h, hipMalloc(&h->d_keep, np));   // + one read-ahead group
        G_TRY(h, hipMalloc(&h->d_key_b, 8 * np * VOX_BATCH + 64)); G_TRY(h, hipMemset(h->d_key_b, 0, 8 * np * VOX_BATCH + 64)); G_TRY(h, hipMalloc(&h->d_keep_b, np * (VOX_BATCH + 1)));
        G_TRY(h, hipMalloc(&h->d_counts, sizeof(int) * VOX_BATCH)); G_TRY(h, hipHostMalloc(&h->h_counts, sizeof(int) * VOX_BATCH));
        G_TRY(h, hipMalloc(&h->d_cs, 8 * (size_t)max_candidates));
        G_TRY(h, hipMalloc(&h->d_idx, 8 * (np * 1024 + 64)));                // up to 1024 rotated scans of max_points points (+ read-ahead pad)
        G_TRY(h, hipMemset(h->d_idx, 0, 8 * (np * 1024 + 64)));
        G_TRY(h, hipMalloc(&h->d_cells, 2 * (size_t)max_cells)); G_TRY(h, hipMalloc(&h->d_cells2, 2 * (size_t)max_cells));
        G_TRY(h, hipMalloc(&h->d_hit, 2 * 32768)); G_TRY(h, hipMalloc(&h->d_miss, 2 * 32768));
        G_TRY(h, hipMalloc(&h->d_mis, 8 * np)); G_TRY(h, hipMalloc(&h->d_ends, 8 * (2 * np + 1))); G_TRY(h, hipMalloc(&h->d_bad, sizeof(int)));
        G_TRY(h, hipMalloc(&h->d_bb, sizeof(BestRec) * 1024));                   // one record per rotated scan
        G_TRY(h, hipMalloc(&h->d_best, sizeof(BestRec)));
        G_TRY(h, hipMalloc(&h->d_count, sizeof(int)));
        G_TRY(h, hipHostMalloc(&h->h_pts, 8 * np)); G_TRY(h, hipHostMalloc(&h->h_count, sizeof(int)));
        G_TRY(h, hipHostMalloc(&h->h_best, sizeof(BestRec)));
        G_TRY(h, hipMalloc(&h->d_box, 4 * sizeof(int))); G_TRY(h, hipHostMalloc(&h->h_box, 4 * sizeof(int)));
        G_TRY(h, hipMalloc(&h->d_refine, sizeof(RefineOut))); G_TRY(h, hipHostMalloc(&h->h_refine, sizeof(RefineOut)));
        return RGRID_OK;
    }();
    if (rc != RGRID_OK) { std::fprintf(stderr, "rgrid_create: %s\n", h->hip_error.c_str()); rgrid_destroy(h); return rc; }
    *out = h;
    return RGRID_OK;
}

void rgrid_destroy(rgrid_t *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void *ptrs[] = {h->d_in, h->d_a, h->d_b, h->d_cs, h->d_key, h->d_idx, h->d_keep, h->d_cells, h->d_bb, h->d_best, h->d_count,
                    h->d_hit, h->d_miss, h->d_mis, h->d_ends, h->d_bad, h->d_cells2, h->d_refine, h->d_tex, h->d_box, h->d_key_b, h->d_keep_b, h->d_counts};
    for (void *p : ptrs) (void)hipFree(p);
    if (h->h_pts) (void)hipHostFree(h->h_pts);
    if (h->h_count) (void)hipHostFree(h->h_count);
    if (h->h_best) (void)hipHostFree(h->h_best);
    if (h->h_refine) (void)hipHostFree(h->h_refine);
    if (h->h_box) (void)hipHostFree(h->h_box);
    if (h->h_counts) (void)hipHostFree(h->h_counts);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int rgrid_voxel_filter(rgrid_t *h, const float *xy, int n, float resolution, float *out_xy, int out_cap, int *m)
{
    if (!h || !m || n < 0 || (n > 0 && !xy) || !(resolution > 0.f) || out_cap < 0 || (out_cap > 0 && !out_xy)) return RGRID_ERR_INVALID;
    *m = 0;
    if (n == 0) return RGRID_OK;
    if (n > h->max_points) return RGRID_ERR_CAPACITY;
    G_TRY(h, hipSetDevice(h->device));
    int rc = upload_points(h, xy, n);
    if (rc != RGRID_OK) return rc;
    int cnt = 0;
    rc = voxel_pass(h, h->d_in, n, resolution, h->d_a, &cnt);
    if (rc != RGRID_OK) return rc;
    rc = download_points(h, h->d_a, cnt, out_xy, out_cap);
    if (rc != RGRID_OK) return rc;
    *m = cnt;
    return RGRID_OK;
}

int rgrid_adaptive_voxel_filter(rgrid_t *h, const float *xy, int n, double max_length, double min_num_points,
                                double max_range, float *out_xy, int out_cap, int *m)
{
    if (!h || !m || n < 0 || (n > 0 && !xy) || !(max_length > 0.) || out_cap < 0 || (out_cap > 0 && !out_xy)) return RGRID_ERR_INVALID;
    *m = 0;
    if (n == 0) return RGRID_OK;
    if (n > h->max_points) return RGRID_ERR_CAPACITY;
    G_TRY(h, hipSetDevice(h->device));
    int rc = upload_points(h, xy, n);
    if (rc != RGRID_OK) return rc;
    // FilterByMaxRange (voxel_filter.cc:15-27) -> d_key is free here, the gated cloud goes to d_in's twin: reuse d_b as "in"
    hipLaunchKernelGGL(kg_range_gate, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d_in, n, (float)max_range, h->d_keep);
    hipLaunchKernelGGL(kg_compact, dim3(1), dim3(1024), 0, h->stream, h->d_in, h->d_keep, n, h->d_b, h->d_count);
    G_TRY(h, hipMemcpyAsync(h->h_count, h->d_count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    G_TRY(h, hipStreamSynchronize(h->stream));
    const int ni = *h->h_count;
    const float *src = h->d_b;                 // the gated cloud
    int cnt = ni;
    const float *final_buf = src;
    // AdaptivelyVoxelFiltered (voxel_filter.cc:29-76).  Its decisions only need the survivor COUNT of a voxel size, and
    // the sizes it can ask for are known in advance: the halving ladder (:47-48) and, between the bracketing pair, the
    // bisection tree (:57-70).  So the ladder is one batched launch and the tree is evaluated a few levels at a time
    // (every node of those levels at once), the host walks the answers: 3-4 round trips instead of a dozen, and the
    // point set is compacted once, for the size the search ends on.
    if (!((double)ni <= min_num_points)) {                                         // :33-37
        const float maxl = (float)max_length;
        float ladder[VOX_BATCH];
        int counts[VOX_BATCH];
        int R = 0;
        ladder[R++] = maxl;                                                        // :38
        for (float high = maxl; high > 1e-2f * maxl && R < VOX_BATCH; high /= 2.f) ladder[R++] = high / 2.f;   // :47-50
        rc = voxel_counts(h, src, ni, ladder, R, counts);
        if (rc != RGRID_OK) return rc;
        int slot = R - 1;                                                          // nothing dense enough: the last size tried (:75)
        if ((double)counts[0] >= min_num_points) slot = 0;                         // :39-43
        else {
            int k = 1;
            while (k < R && !((double)counts[k] >= min_num_points)) ++k;
            if (k < R) {
                slot = k;
                float low = ladder[k], high = (k == 1) ? maxl : ladder[k - 1];     // high of that iteration = the previous low
                unsigned char *park = h->d_keep_b + (size_t)VOX_BATCH * ni;        // `result` while later batches reuse the slices
                bool parked = false;
                const int depth = ni <= 4096 ? 5 : (ni <= 8192 ? 3 : 2);
                while ((high - low) / low > 1e-1f) {                               // :57
                    // the next `depth` levels of the bisection tree under (low, high), breadth first
                    struct Node { float low, high, mid; int yes, no; };
                    Node nodes[VOX_BATCH];
                    int nn = 0, level_begin = 0;
                    nodes[nn++] = Node{low, high, (low + high) / 2.f, -1, -1};
                    for (int lv = 1; lv < depth; ++lv) {
                        const int level_end = nn;
                        for (int q = level_begin; q < level_end; ++q) {
                            const Node nd = nodes[q];
                            if ((nd.high - nd.mid) / nd.mid > 1e-1f && nn < VOX_BATCH) {          // count(mid) >= min: low = mid
                                nodes[q].yes = nn; nodes[nn++] = Node{nd.mid, nd.high, (nd.mid + nd.high) / 2.f, -1, -1};
                            }
                            if ((nd.mid - nd.low) / nd.low > 1e-1f && nn < VOX_BATCH) {            // else: high = mid
                                nodes[q].no = nn; nodes[nn++] = Node{nd.low, nd.mid, (nd.low + nd.mid) / 2.f, -1, -1};
                            }
                        }
                        level_begin = level_end;
                    }
                    float mids[VOX_BATCH];
                    for (int q = 0; q < nn; ++q) mids[q] = nodes[q].mid;
                    if (!parked) {                                                 // `result` lives in a slice the batch overwrites
                        G_TRY(h, hipMemcpyAsync(park, h->d_keep_b + (size_t)slot * ni, (size_t)ni, hipMemcpyDeviceToDevice, h->stream));
                        parked = true;
                    }
                    rc = voxel_counts(h, src, ni, mids, nn, counts);
                    if (rc != RGRID_OK) return rc;
                    int q = 0, chosen = -1;
                    while (q >= 0) {
                        const bool dense = (double)counts[q] >= min_num_points;
                        if (dense) { low = nodes[q].mid; chosen = q; } else high = nodes[q].mid;
                        q = dense ? nodes[q].yes : nodes[q].no;                    // -1: the loop condition failed there, or the depth ran out
                    }
                    if (chosen >= 0) {                                             // result = candidate (:63-66)
                        slot = chosen; parked = false;
                    }
                }
                if (parked) slot = VOX_BATCH;
            }
        }
        hipLaunchKernelGGL(kg_compact, dim3(1), dim3(1024), 0, h->stream, src, h->d_keep_b + (size_t)slot * ni, ni, h->d_a, h->d_count);
        G_TRY(h, hipMemcpyAsync(h->h_count, h->d_count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        G_TRY(h, hipStreamSynchronize(h->stream));
        cnt = *h->h_count;
        final_buf = h->d_a;
    }
    rc = download_points(h, final_buf, cnt, out_xy, out_cap);
    if (rc != RGRID_OK) return rc;
    *m = cnt;
    return RGRID_OK;
}

int rgrid_set_grid(rgrid_t *h, const uint16_t *cells, int num_x_cells, int num_y_cells, double resolution,
                   double max_x, double max_y)
{
    if (!h || !cells || num_x_cells < 1 || num_y_cells < 1 || !(resolution > 0.)) return RGRID_ERR_INVALID;
    if ((long long)num_x_cells * num_y_cells > h->max_cells) return RGRID_ERR_CAPACITY;
    G_TRY(h, hipSetDevice(h->device));
    G_TRY(h, hipMemcpyAsync(h->d_cells, cells, sizeof(uint16_t) * (size_t)num_x_cells * num_y_cells, hipMemcpyHostToDevice, h->stream));
    G_TRY(h, hipStreamSynchronize(h->stream));
    h->nx = num_x_cells; h->ny = num_y_cells; h->resolution = resolution; h->max_x = max_x; h->max_y = max_y;
    h->have_grid = true;
    return RGRID_OK;
}

int rgrid_insert(rgrid_t *h, const float origin_xy[2], const float *returns_xy, int n_returns, const float *misses_xy,
                 int n_misses, float hit_probability, float miss_probability, int insert_free_space)
{
    if (!h || !origin_xy || n_returns < 0 || n_misses < 0 || (n_returns > 0 && !returns_xy) || (n_misses > 0 && !misses_xy))
        return RGRID_ERR_INVALID;
    if (!h->have_grid || !(hit_probability > 0.f && hit_probability < 1.f) || !(miss_probability > 0.f && miss_probability < 1.f))
        return RGRID_ERR_INVALID;
    if (n_returns > h->max_points || n_misses > h->max_points) return RGRID_ERR_CAPACITY;
    G_TRY(h, hipSetDevice(h->device));
    if (hit_probability != h->tab_hit_p || miss_probability != h->tab_miss_p) {      // the tables only depend on the two options
        std::vector<unsigned short> t(32768);
        lookup_table(hit_probability, t.data());
        G_TRY(h, hipMemcpy(h->d_hit, t.data(), 2 * 32768, hipMemcpyHostToDevice));
        lookup_table(miss_probability, t.data());
        G_TRY(h, hipMemcpy(h->d_miss, t.data(), 2 * 32768, hipMemcpyHostToDevice));
        h->tab_hit_p = hit_probability; h->tab_miss_p = miss_probability;
    }
    if (n_returns > 0) {
        std::memcpy(h->h_pts, returns_xy, sizeof(float) * 2 * (size_t)n_returns);
        G_TRY(h, hipMemcpyAsync(h->d_in, h->h_pts, sizeof(float) * 2 * (size_t)n_returns, hipMemcpyHostToDevice, h->stream));
        G_TRY(h, hipStreamSynchronize(h->stream));                                   // h_pts is reused for the misses below
    }
    if (n_misses > 0) {
        std::memcpy(h->h_pts, misses_xy, sizeof(float) * 2 * (size_t)n_misses);
        G_TRY(h, hipMemcpyAsync(h->d_mis, h->h_pts, sizeof(float) * 2 * (size_t)n_misses, hipMemcpyHostToDevice, h->stream));
    }
    G_TRY(h, hipMemsetAsync(h->d_bad, 0, sizeof(int), h->stream));
    InsertArgs A;
    A.nx = h->nx; A.ny = h->ny; A.n_ret = n_returns; A.n_miss = n_misses;
    A.max_x = h->max_x; A.max_y = h->max_y; A.rs = h->resolution / SUBPX;
    A.ox = origin_xy[0]; A.oy = origin_xy[1];
    const int n = n_returns + n_misses;
    hipLaunchKernelGGL(kg_ends, dim3((n + 1 + 255) / 256), dim3(256), 0, h->stream, A, h->d_in, h->d_mis, h->d_ends, h->d_bad);
    if (n_returns > 0)
        hipLaunchKernelGGL(kg_hits, dim3((n_returns + 255) / 256), dim3(256), 0, h->stream, A, h->d_ends, h->d_bad, h->d_cells, h->d_hit);
    if (insert_free_space && n > 0)
        hipLaunchKernelGGL(kg_rays, dim3((n + 3) / 4), dim3(256), 0, h->stream, A, h->d_ends, h->d_bad, h->d_cells, h->d_miss);
    const long long ncells = (long long)h->nx * h->ny;
    hipLaunchKernelGGL(kg_finish, dim3((unsigned)((ncells + 255) / 256)), dim3(256), 0, h->stream, h->d_cells, ncells, h->d_bad);
    G_TRY(h, hipMemcpyAsync(h->h_count, h->d_bad, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    G_TRY(h, hipStreamSynchronize(h->stream));
    return *h->h_count ? RGRID_ERR_CAPACITY : RGRID_OK;
}

namespace {
// Grid2D::GrowLimits(point) on the limits only (grid_2d.cc:64-75,93); false if the grown grid exceeds `max_cells`
bool grow_limits_for(float px, float py, double res, long long max_cells, int &nx, int &ny, double &max_x, double &max_y, int &off_x, int &off_y)
{
#pragma clang fp contract(off)
    for (;;) {
        const long ix = std::lround((max_y - (double)py) / res - 0.5), iy = std::lround((max_x - (double)px) / res - 0.5);
        if (ix >= 0 && iy >= 0 && ix < nx && iy < ny) return true;
        if (4ll * nx * ny > max_cells) return false;
        const int xo = nx / 2, yo = ny / 2;
        max_x = max_x + res * (double)yo;
        max_y = max_y + res * (double)xo;
        nx *= 2; ny *= 2; off_x += xo; off_y += yo;
    }
}
}  // namespace

int rgrid_grow_as_needed(rgrid_t *h, const float origin_xy[2], const float *returns_xy, int n_returns, const float *misses_xy,
                         int n_misses)
{
    if (!h || !origin_xy || n_returns < 0 || n_misses < 0 || (n_returns > 0 && !returns_xy) || (n_misses > 0 && !misses_xy))
        return RGRID_ERR_INVALID;
    if (!h->have_grid) return RGRID_ERR_INVALID;
    // Eigen::AlignedBox2f(origin).extend(every return and miss)  (probability_grid_range_data_inserter_2d.cc:23-33)
    float lo[2] = {origin_xy[0], origin_xy[1]}, hi[2] = {origin_xy[0], origin_xy[1]};
    auto extend = [&](const float *p, int n) {
        for (int i = 0; i < 2 * n; ++i) {
            const float v = p[i];
            if (!std::isfinite(v)) return false;
            if (v < lo[i & 1]) lo[i & 1] = v;
            if (v > hi[i & 1]) hi[i & 1] = v;
        }
        return true;
    };
    if (!std::isfinite(lo[0]) || !std::isfinite(lo[1]) || !extend(returns_xy, n_returns) || !extend(misses_xy, n_misses)) return RGRID_ERR_INVALID;
    const float pad = 1e-6f;                                                        // kPadding (:25)
    int nx = h->nx, ny = h->ny, off_x = 0, off_y = 0;
    double max_x = h->max_x, max_y = h->max_y;
    if (!grow_limits_for(lo[0] - pad, lo[1] - pad, h->resolution, h->max_cells, nx, ny, max_x, max_y, off_x, off_y) ||
        !grow_limits_for(hi[0] + pad, hi[1] + pad, h->resolution, h->max_cells, nx, ny, max_x, max_y, off_x, off_y))
        return RGRID_ERR_CAPACITY;
    if (nx == h->nx && ny == h->ny) return RGRID_OK;
    G_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(kg_grow, dim3((nx + 255) / 256, ny), dim3(256), 0, h->stream, h->d_cells, h->nx, h->ny, h->d_cells2, nx, ny, off_x, off_y);
    G_TRY(h, hipGetLastError());
    G_TRY(h, hipStreamSynchronize(h->stream));
    std::swap(h->d_cells, h->d_cells2);
    h->nx = nx; h->ny = ny; h->max_x = max_x; h->max_y = max_y;
    return RGRID_OK;
}

int rgrid_get_limits(rgrid_t *h, int *num_x_cells, int *num_y_cells, double *resolution, double *max_x, double *max_y)
{
    if (!h || !h->have_grid) return RGRID_ERR_INVALID;
    if (num_x_cells) *num_x_cells = h->nx;
    if (num_y_cells) *num_y_cells = h->ny;
    if (resolution) *resolution = h->resolution;
    if (max_x) *max_x = h->max_x;
    if (max_y) *max_y = h->max_y;
    return RGRID_OK;
}

int rgrid_get_grid(rgrid_t *h, uint16_t *cells, long cap)
{
    if (!h || !cells || !h->have_grid) return RGRID_ERR_INVALID;
    const long long ncells = (long long)h->nx * h->ny;
    if (cap < ncells) return RGRID_ERR_BUFFER;
    G_TRY(h, hipSetDevice(h->device));
    G_TRY(h, hipMemcpyAsync(cells, h->d_cells, sizeof(uint16_t) * (size_t)ncells, hipMemcpyDeviceToHost, h->stream));
    G_TRY(h, hipStreamSynchronize(h->stream));
    return RGRID_OK;
}

int rgrid_match(rgrid_t *h, const rgrid_match_options *opt, const double initial_pose[3], const float *points_xy,
                int n, double pose_estimate[3], double *score, int best3[3], int info3[3])
{
#pragma clang fp contract(off)
    if (!h || !opt || !initial_pose || !pose_estimate || !score || n < 0 || (n > 0 && !points_xy)) return RGRID_ERR_INVALID;
    if (!h->have_grid) return RGRID_ERR_INVALID;
    if (n == 0) return RGRID_ERR_EMPTY;
    if (n > h->max_points) return RGRID_ERR_CAPACITY;
    G_TRY(h, hipSetDevice(h->device));
    // initial rotation of the cloud (real_time_correlative_scan_matcher_2d.cc:91-97), host float32
    float c0, s0;
    rotation_cs((float)initial_pose[2], &c0, &s0);
    float max_scan_range = 3.f * (float)h->resolution;                                   // correlative_scan_matcher_2d.cc:18-24
    for (int i = 0; i < n; ++i) {
        const float x = points_xy[2 * i], y = points_xy[2 * i + 1];
        const float rx = c0 * x - s0 * y, ry = s0 * x + c0 * y;
        h->h_pts[2 * i] = rx; h->h_pts[2 * i + 1] = ry;
        const float range = std::sqrt(rx * rx + ry * ry);
        if (range > max_scan_range) max_scan_range = range;
    }
    const double res = h->resolution;
    const double kSafetyMargin = 1. - 1e-3;
    const double step = kSafetyMargin * std::acos(1. - (res * res) / (2. * (double)(max_scan_range * max_scan_range)));   // :25-28
    const int num_angular = (int)std::ceil(opt->angular_search_window / step);            // :29-31
    const int num_scans = 2 * num_angular + 1;
    const int num_linear = (int)std::ceil(opt->linear_search_window / res);               // :33-34
    const long long W = 2LL * num_linear + 1, ncand = (long long)num_scans * W * W;
    if (num_scans > 1024 || ncand > h->max_candidates) return RGRID_ERR_CAPACITY;
    std::vector<float> cs(2 * (size_t)num_scans);
    double delta_theta = -num_angular * step;                                             // :90-94 (accumulated in double)
    for (int s = 0; s < num_scans; ++s, delta_theta += step) rotation_cs((float)delta_theta, &cs[2 * s], &cs[2 * s + 1]);
    G_TRY(h, hipMemcpyAsync(h->d_in, h->h_pts, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, h->stream));
    G_TRY(h, hipMemcpyAsync(h->d_cs, cs.data(), sizeof(float) * cs.size(), hipMemcpyHostToDevice, h->stream));
    MatchArgs A;
    A.n = n; A.num_scans = num_scans; A.num_linear = num_linear; A.nx = h->nx; A.ny = h->ny;
    A.tx = (float)initial_pose[0]; A.ty = (float)initial_pose[1];
    A.resolution = res; A.max_x = h->max_x; A.max_y = h->max_y;
    A.num_angular_d = (double)num_angular; A.step = step;
    A.wt = opt->translation_delta_cost_weight; A.wr = opt->rotation_delta_cost_weight;
    const int nb_d = (num_scans * n + 255) / 256, nb_s = num_scans;
    hipLaunchKernelGGL(kg_discretize, dim3(nb_d), dim3(256), 0, h->stream, A, h->d_in, h->d_cs, h->d_idx);
    hipLaunchKernelGGL(kg_score, dim3(nb_s), dim3(128), 0, h->stream, A, h->d_idx, h->d_cells, h->d_bb);
    hipLaunchKernelGGL(kg_best, dim3(1), dim3(256), 0, h->stream, h->d_bb, nb_s, h->d_best);
    G_TRY(h, hipMemcpyAsync(h->h_best, h->d_best, sizeof(BestRec), hipMemcpyDeviceToHost, h->stream));
    G_TRY(h, hipStreamSynchronize(h->stream));                                            // cs stays alive until here
    const int id = h->h_best->id;
    const int scan = id / (int)(W * W), r = id - scan * (int)(W * W);
    const int xo = r / (int)W - num_linear, yo = r - (r / (int)W) * (int)W - num_linear;
    const double x = -yo * res, y = -xo * res, orientation = (scan - num_angular) * step;
    pose_estimate[0] = initial_pose[0] + x;                                               // :106-110
    pose_estimate[1] = initial_pose[1] + y;
    pose_estimate[2] = initial_pose[2] + orientation;
    *score = (double)h->h_best->score;
    if (best3) { best3[0] = scan; best3[1] = xo; best3[2] = yo; }
    if (info3) { info3[0] = num_scans; info3[1] = num_linear; info3[2] = (int)ncand; }
    return RGRID_OK;
}

// ---- mapping::MapBuilder::AddRangeData (src/mapping/map_builder.cc:57-108) as one call ------------------------------
// Host side = what the reference's host does between the steps: a few rigid transforms (its float32 round trips
// through Eigen quaternions restated operation by operation, as in reflector_ekf_slam_amd/map_builder.py) and the
// decisions; every per-point / per-cell step is one of the entry points above.
namespace {
float yaw_of_quaternion_f32(float w, float z)          // transform::GetYaw(Quaternionf(w,0,0,z)): q * UnitX = (1 - z 2z, w 2z, 0)
{
#pragma clang fp contract(off)
    const float two_z = z + z;
    const float y = w * two_z, x = 1.f - z * two_z;
    return (float)std::atan2((double)y, (double)x);
}
double yaw_of_quaternion_f64(double w, double z)
{
#pragma clang fp contract(off)
    const double two_z = z + z;
    return std::atan2(w * two_z, 1.0 - z * two_z);
}
void rigid2f_apply(float tx, float ty, float yaw, const float *in, int n, float *out)   // Rigid2f(t, Rotation2Df(yaw)) * p
{
#pragma clang fp contract(off)
    const float c = (float)std::cos((double)yaw), s = (float)std::sin((double)yaw);
    for (int i = 0; i < n; ++i) {
        const float x = in[2 * i], y = in[2 * i + 1];
        out[2 * i] = (c * x - s * y) + tx;
        out[2 * i + 1] = (s * x + c * y) + ty;
    }
}
}  // namespace

int rgrid_add_range_data(rgrid_t *h, const rgrid_map_builder_options *opt, const float origin_xy[2], const float *returns_xy,
                         int n_returns, const float *misses_xy, int n_misses, const double ekf_pose[3], double local_pose[3],
                         float *returns_in_local, int *status)
{
#pragma clang fp contract(off)
    if (!h || !opt || !origin_xy || !ekf_pose || !local_pose || n_returns < 0 || n_misses < 0 || (n_returns > 0 && !returns_xy) ||
        (n_misses > 0 && !misses_xy))
        return RGRID_ERR_INVALID;
    if (status) *status = RGRID_SCAN_DROPPED_EMPTY;
    if (n_returns == 0) return RGRID_OK;                                          // "Dropped empty horizontal range data." (:63-67)
    if (n_returns > h->max_points || n_misses > h->max_points) return RGRID_ERR_CAPACITY;
    const double x = ekf_pose[0], y = ekf_pose[1], theta = ekf_pose[2];
    const double qw = std::cos(theta / 2), qz = std::sin(theta / 2);             // the caller's quaternion (src/ros_node.cc:548)
    // TransformToGravityAlignedFrameAndFilter (:20-32): Rotation(q).cast<float>() -> Project2D -> Rigid2f(0, GetYaw)
    const float yaw_g = yaw_of_quaternion_f32((float)qw, (float)qz);
    std::vector<float> ret((size_t)2 * n_returns), mis((size_t)2 * (n_misses > 0 ? n_misses : 1)), fr((size_t)2 * n_returns),
        fm((size_t)2 * (n_misses > 0 ? n_misses : 1)), av((size_t)2 * n_returns);
    float org[2];
    rigid2f_apply(0.f, 0.f, yaw_g, origin_xy, 1, org);
    rigid2f_apply(0.f, 0.f, yaw_g, returns_xy, n_returns, ret.data());
    if (n_misses > 0) rigid2f_apply(0.f, 0.f, yaw_g, misses_xy, n_misses, mis.data());
    int nfr = 0, nfm = 0, nav = 0;
    int rc = rgrid_voxel_filter(h, ret.data(), n_returns, opt->voxel_filter_size, fr.data(), n_returns, &nfr);
    if (rc != RGRID_OK) return rc;
    if (n_misses > 0) { rc = rgrid_voxel_filter(h, mis.data(), n_misses, opt->voxel_filter_size, fm.data(), n_misses, &nfm); if (rc != RGRID_OK) return rc; }
    // pose_prediction = Project2D(ekf_pose * gravity_alignment.inverse()): the rotations cancel (:70-71)
    const double prediction[3] = {x, y, 0.0};
    rc = rgrid_adaptive_voxel_filter(h, fr.data(), nfr, opt->adaptive_max_length, opt->adaptive_min_num_points, opt->adaptive_max_range,
                                     av.data(), n_returns, &nav);
    if (rc != RGRID_OK) return rc;
    if (status) *status = RGRID_SCAN_FILTERED_EMPTY;
    if (nav == 0) return RGRID_OK;                                               // (:74-77)
    double est[3] = {prediction[0], prediction[1], prediction[2]};               // ScanMatch (:34-55): no submap yet -> the prediction
    if (h->have_grid) {
        double coarse[3], score = 0;
        rc = rgrid_match(h, &opt->match, prediction, av.data(), nav, coarse, &score, nullptr, nullptr);
        if (rc != RGRID_OK) return rc;
        rc = rgrid_refine_match(h, &opt->refine, prediction, coarse, av.data(), nav, est, nullptr);
        if (rc != RGRID_OK) return rc;
    }
    // pose_estimate = Embed3D(pose_estimate_2d) * gravity_alignment: quaternion product about z, in double (:86-87)
    const double aw = std::cos(0.5 * est[2]), az = std::sin(0.5 * est[2]);
    const double pw = aw * qw - az * qz, pz = aw * qz + az * qw;
    local_pose[0] = est[0]; local_pose[1] = est[1]; local_pose[2] = yaw_of_quaternion_f64(pw, pz);
    if (returns_in_local)                                                         // range_data_in_local (:89-90), the raw returns
        rigid2f_apply((float)est[0], (float)est[1], yaw_of_quaternion_f32((float)pw, (float)pz), returns_xy, n_returns, returns_in_local);
    // range_data_in_local2 = TransformRangeData(gravity-aligned filtered data, Embed3D(pose_estimate_2d.cast<float>())) (:92-94)
    const float af = (float)est[2], ha = 0.5f * af;
    const float yaw2 = yaw_of_quaternion_f32((float)std::cos((double)ha), (float)std::sin((double)ha));
    float org2[2];
    rigid2f_apply((float)est[0], (float)est[1], yaw2, org, 1, org2);
    rigid2f_apply((float)est[0], (float)est[1], yaw2, fr.data(), nfr, ret.data());
    if (nfm > 0) rigid2f_apply((float)est[0], (float)est[1], yaw2, fm.data(), nfm, mis.data());
    if (!h->have_grid) {                                                          // InsertIntoSubmap / CreateGrid (:110-126)
        const int n0 = 100;                                                      // kInitialSubmapSize
        if ((long long)n0 * n0 > h->max_cells) return RGRID_ERR_CAPACITY;
        const double resolution = (double)opt->resolution;                       // `float resolution = options_.resolution`
        const double half = 0.5 * n0 * resolution;
        G_TRY(h, hipSetDevice(h->device));
        G_TRY(h, hipMemsetAsync(h->d_cells, 0, sizeof(uint16_t) * (size_t)n0 * n0, h->stream));
        G_TRY(h, hipStreamSynchronize(h->stream));
        h->nx = n0; h->ny = n0; h->resolution = resolution; h->max_x = (double)org2[0] + half; h->max_y = (double)org2[1] + half;
        h->have_grid = true;
    }
    rc = rgrid_grow_as_needed(h, org2, ret.data(), nfr, nfm > 0 ? mis.data() : nullptr, nfm);
    if (rc != RGRID_OK) return rc;
    rc = rgrid_insert(h, org2, ret.data(), nfr, nfm > 0 ? mis.data() : nullptr, nfm, opt->hit_probability, opt->miss_probability,
                      opt->insert_free_space);
    if (rc != RGRID_OK) return rc;
    if (status) *status = RGRID_SCAN_INSERTED;
    return RGRID_OK;
}

}  // extern "C"
