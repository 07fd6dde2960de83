// ekf_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the EKF hot path.
//
// One HandleObservationMessage (reference reflector_ekf_slam.cc:229-368) is the arithmetic
//     front end -> mid -> downdate (-> augment)
// on the handle's stream, with every size (n, m, match lists) resident in HBM (RekfCtl) so the host never waits for
// the device between scans.  Launches (round 3, rekf_api.hip "lazy downdate"): a scan's downdate (and augment) are held back and go
// out with the next call; scan after scan that is  k_dd_front (previous downdate + this scan's front end) [-> k_augment] -> k_mid.
//
//   front_role  Predict's pose (cc:154-206) + ReflectorMatch (cc:370-455), one observation per workgroup: k_front_mb on its own, the
//               last workgroups of k_dd_front beside the previous scan's downdate.  Predict's O(n) covariance part is not applied to
//               memory: it travels as (a, b) + predicted pose block (RekfCtl::pred) and k_mid / the downdate apply it to what they read
//   k_mid       ordered compaction, H rows / z - zhat (cc:248-304); W = P H^T and (H P)^T from the <= 5 structural
//               non-zeros of each H row (cc:305,308); S = H W + Q and S^-1 by in-register blocked Gauss-Jordan;
//               K = W S^-1 (FP64 MFMA), mu += K (z - zhat), theta wrap (cc:305-307) -- one launch, the 64 x 64
//               inverse redone by every workgroup rather than handed around
//               (workgroup 0 also: the pose block after the update, RekfCtl::post_C9, and the new reflectors' means, cc:323-342)
//   k_downdate2 / k_dd_front   P += Kn (H P): the FP64 MFMA, LDS-tiled rank-m downdate (cc:308) -- the roofline kernel
//   k_augment   the new landmarks' covariance rows (cc:343-364)
//   k_apply_predict   odometry messages and empty scans are predicted by the HOST (pose mirror, rekf_api.hip) and cost no launch;
//               this kernel applies their composite to P when the device state is needed before the next scan
// Scans with more than 32 matched pairs (or more than 64 observations: k_compact_wide) run the joint update as exact
// block steps, k_mid + k_downdate2 per 32 pairs (see k_mid).
//
// The stored covariance is EXACTLY symmetric, bit for bit: every kernel that writes P writes both halves from ONE computed
// value (P is STORED as its lower triangle, ekf_dev.h: nothing ever reads the other half).  W = P H^T is gathered from the columns of P and (H P)^T(c, r) = W(c, r) is stored from the same values.
// (On a P that is only NEARLY symmetric taking H P := (P H^T)^T is unstable -- the antisymmetric round-off part A then evolves
// as A + (P G) A (G P) instead of the reference's contraction (I - P G) A (I - G P), G = H^T S^-1 H: measured 1e-17 -> 1e-5 in
// 300 scans in round 1 -- which is why nothing here ever leaves the two halves to independent round-off)
//
// Deliberate, stated deviations from the literal Eigen expressions (FP64 round-off
// level, far inside the 1e-5 m parity bar):
//   * S^-1 by Gauss-Jordan without pivoting (S = H P H^T + Q is SPD) instead of
//     Eigen's partial-pivot LU;
//   * more than 32 matched pairs: block-sequential form of the same joint update;
//   * P exactly symmetric (above); consecutive predicts applied to P's landmark rows as their exact composite.
#include "ekf_dev.h"

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

#define WAVE 64

// ----------------------------------------------------------------------------
// small scalar pieces (Motion / motion_terms / corner_predict live in ekf_dev.h: the host's pose mirror evaluates the very
// same source): kept free of FMA contraction so that the float32
// roundings the reference performs (cc:389-393, :431-433, :327-331) see the
// same doubles as a plain x86-64 build of the reference.
// ----------------------------------------------------------------------------
__device__ static void obs_to_global(double x, double y, double c, double s, float px, float py,
                                     float &gx, float &gy)
{
#pragma clang fp contract(off)
    // cc:389-393 / cc:327-331: evaluated in double, rounded to float32 on assignment
    // (c, s = cos, sin of the heading)
    gx = (float)((double)px * c - (double)py * s + x);
    gy = (float)((double)px * s + (double)py * c + y);
}

__device__ static __forceinline__ double yaw_innovation(double delta_theta)
{
#pragma clang fp contract(off)
    // quaternion (w,0,0,z) -> angle-axis z: reference transform.h:46-70 via gps.cc:320-322
    double w = cos(delta_theta / 2), z = sin(delta_theta / 2);
    const double nrm = sqrt(w * w + z * z);
    w /= nrm; z /= nrm;
    if (w < 0.) { w = -w; z = -z; }
    const double angle = 2. * atan2(fabs(z), w);
    const double scale = angle < 1e-7 ? 2. : angle / sin(angle / 2.);
    return scale * z;
}

// bare v_min_f64 / v_max_f64 (fmin/fmax would add a canonicalising v_max x,x per operand)
__device__ static inline double vmin_f64(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ static inline double vmax_f64(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// wave-wide arg-min of (d, j) with the "first minimum in index order" rule
// (cc:414-419 sorts with '<=' and takes front(); ties are UB there).
// one DPP step of the wave-wide (smallest, its first index, second smallest) combine
template <int CTRL, int ROW_MASK> __device__ static inline double dpp_f64(double neutral, double v)
{
    const long long nb = __double_as_longlong(neutral), vb = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp((int)nb, (int)vb, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(nb >> 32), (int)(vb >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ static inline void argmin2_combine(double &a1, int &ja, double &a2, double b1, int jb, double b2);
template <int CTRL, int ROW_MASK> __device__ static inline void argmin2_dpp_step(double &g1, int &gj, double &g2)
{
    const double o1 = dpp_f64<CTRL, ROW_MASK>(1e300, g1), o2 = dpp_f64<CTRL, ROW_MASK>(1e300, g2);
    const int oj = __builtin_amdgcn_update_dpp(-1, gj, CTRL, ROW_MASK, 0xf, false);
    argmin2_combine(g1, gj, g2, o1, oj, o2);
}
__device__ static void wave_argmin(double &d, int &j)
{
    for (int off = 32; off >= 1; off >>= 1) {
        const double od = __shfl_xor(d, off, WAVE);
        const int oj = __shfl_xor(j, off, WAVE);
        const bool take = (oj >= 0) && (j < 0 || od < d || (od == d && oj < j));
        if (take) { d = od; j = oj; }
    }
}

// ----------------------------------------------------------------------------
// k_apply_predict: the covariance part of Predict for predicts the HOST evaluated (odometry messages, cc:208-223, and empty
// scans, cc:235-236, cost no launch: rekf_api.hip advances its mirror of the pose mean and of the 3 x 3 pose block and
// accumulates the composite G = I + a e0 e2^T + b e1 e2^T).  This kernel brings P and mu up to date when somebody needs
// them on the device before the next scan does it in k_front_mb (GetState, PredictState, rekf_reserve): rows / columns 0, 1
// of the landmark part (cc:178 / :202 multiply dense n x n), the pose block and the pose mean by value.  One workgroup.
// ----------------------------------------------------------------------------
#define COV_PF 3              // covariance-predict operands prefetched per thread (covers n <= 3072)
// ---- results for the host without a copy engine: each value is ONE 16-byte system-scope store {double, tag, aux} into
// pinned host memory; the host polls the tags (no hipMemcpy, no wait for the completion signal).
typedef unsigned rekf_u32x4 __attribute__((ext_vector_type(4)));
__device__ static void host_slot_store(RekfHostSlot *p, double v, int seq, int aux)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const rekf_u32x4 w = {(unsigned)b, (unsigned)(b >> 32), (unsigned)seq, (unsigned)aux};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");     // (s_nop: see dd_store_sc1)
}

// write-through (agent scope) stores for what the NEXT kernel reads from other XCDs: see DD_STORE at k_downdate2
__device__ static inline void store_wt(double *p, double v)
{
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
typedef double rekf_v2d __attribute__((ext_vector_type(2)));
__device__ static inline void store_wt2(double *p, rekf_v2d v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// ----------------------------------------------------------------------------
// MATCH GRID (ekf_dev.h, RekfCtl::grid_state): binning of one landmark, the rebuild kernel.
// ----------------------------------------------------------------------------
__device__ static inline void grid_insert(const RekfDev &d, RekfCtl *ctl, int j, float fx, float fy)
{
    const int h = rekf_grid_hash((int)floorf(fx), (int)floorf(fy), d.grid_mask);
    const int slot = atomicAdd(&rekf_grid_cnt(d)[h], 1);
    int e = -1;
    if (slot < REKF_GRID_SLOTS) {
        e = REKF_GRID_SLOTS * h + slot;
        rekf_grid_id(d)[e] = j;
        float *x0 = rekf_grid_xy(d, 0), *x1 = rekf_grid_xy(d, 1);         // (both halves: whichever the next launch starts from)
        x0[2 * e] = fx; x0[2 * e + 1] = fy; x1[2 * e] = fx; x1[2 * e + 1] = fy;
    } else {
        __hip_atomic_store(&ctl->grid_overflow, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ctl->grid_state, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    rekf_grid_slot(d)[j] = e;
    d.grid_p0[2 * j] = fx; d.grid_p0[2 * j + 1] = fy;
}
// One workgroup: empty the table, bin every landmark of the current mean where it stands now.  The host launches it (in stream order, in
// front of the scan that wants the grid) after rekf_create / rekf_set_state / rekf_reserve and when a kernel has reported drift.
__global__ __launch_bounds__(1024) void k_grid_build(RekfDev d, int n, int note_tag)
{
    RekfCtl *ctl = d.ctl;
    const int tid = threadIdx.x;
    const int nb = d.grid_mask + 1;
    if (tid == 0) { ctl->grid_overflow = 0; ctl->grid_state = 1; }
    for (int e = tid; e < nb; e += 1024) rekf_grid_cnt(d)[e] = 0;
    __threadfence_block();
    __syncthreads();
    const int L = (n - 3) / 2;
    for (int j = tid; j < L; j += 1024) grid_insert(d, ctl, j, (float)d.mu[3 + 2 * j], (float)d.mu[4 + 2 * j]);
    __threadfence_block();
    __syncthreads();
    // (a table too small for this world: tell the host, which stops asking for the grid)
    if (tid == 0 && d.grid_note && __hip_atomic_load(&ctl->grid_overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) host_slot_store(d.grid_note, 2.0, note_tag, 0);
}
void rekf_launch_grid_build(const RekfDev &d, int n, int note_tag, hipStream_t s)
{
    hipLaunchKernelGGL(k_grid_build, dim3(1), dim3(1024), 0, s, d, n, note_tag);
}

__global__ __launch_bounds__(1024) void k_apply_predict(RekfDev d, RekfFrontArgs A)
{
#pragma clang fp contract(off)
    const int tid = threadIdx.x;
    RekfCtl *ctl = d.ctl;
    double *__restrict__ P = d.P;
    const size_t ld = (size_t)d.ld;
    const int n = (d.n_known >= 0) ? d.n_known : ctl->n;
    // P is stored as its lower triangle (ekf_dev.h): Predict touches the coalesced columns P(idx, 0..1) against P(idx, 2); the rows
    // P(0..1, idx) are the same numbers and exist nowhere else
    const double a = A.pre_ab[0], b = A.pre_ab[1];
    double c0[COV_PF], c1[COV_PF], c2[COV_PF];
#pragma unroll
    for (int t = 0; t < COV_PF; ++t) {
        const int idx = tid + 1024 * t;
        if (idx >= 3 && idx < n) { c0[t] = P[idx + 0 * ld]; c1[t] = P[idx + 1 * ld]; c2[t] = P[idx + 2 * ld]; }
    }
#pragma unroll
    for (int t = 0; t < COV_PF; ++t) {
        const int idx = tid + 1024 * t;
        if (idx >= 3 && idx < n) {
            const double n0 = c0[t] + a * c2[t], n1 = c1[t] + b * c2[t];
            P[idx + 0 * ld] = n0;
            P[idx + 1 * ld] = n1;
        }
    }
    for (int idx = tid + 1024 * COV_PF; idx < n; idx += 1024) {
        const double p2 = P[idx + 2 * ld];
        const double n0 = P[idx + 0 * ld] + a * p2, n1 = P[idx + 1 * ld] + b * p2;
        P[idx + 0 * ld] = n0;
        P[idx + 1 * ld] = n1;
    }
    if (tid < 9) P[(tid % 3) + (size_t)(tid / 3) * ld] = A.pre_C9[0 + tid];
    if (tid >= 64 && tid < 67) d.mu[tid - 64] = A.pre_pose[tid - 64];
}

// ----------------------------------------------------------------------------
// k_front_mb: the observation path's front end spread over FRONT_MB workgroups.
//
// A single-workgroup front kernel is bounded by one CU's VALU and by serial reductions (15 us at
// L = 1024, K = 32).  Here every workgroup recomputes the (cheap) predicted pose from the OLD
// mean -- nobody writes mu[0..2] in this kernel: the predicted pose goes to ctl->pose_pred and is
// committed by k_gain together with the update -- then takes a 1/FRONT_MB slice of the
// covariance predict and whole observations of ReflectorMatch (all 16 waves sweep disjoint
// landmark slices, wave-wide literal arg-min).  The ordered compaction and the H rows (the tail of
// that kernel) are resolved by k_mid from the per-observation results left in ctl->obs_kind/obs_idx.
// ----------------------------------------------------------------------------
#define FRONT_MB 32

// arg-min that also carries the runner-up value: (v, j) = smallest value / its first index, v2 = the
// second smallest value over all candidates (the lanes' own runner-ups included)
__device__ static inline void argmin2_combine(double &a1, int &ja, double &a2, double b1, int jb, double b2)
{
    const bool take = (jb >= 0) && (ja < 0 || b1 < a1 || (b1 == a1 && jb < ja));
    const double lose = take ? a1 : b1;                 // the larger of the two minima
    const double n2 = vmin_f64(vmin_f64(a2, b2), (ja >= 0 && jb >= 0) ? lose : 1e300);
    if (take) { a1 = b1; ja = jb; }
    a2 = n2;
}
// One wave: the scan's per-observation match results (kind, landmark / map index; lane = observation, K <= 32) -> the record k_mid
// works with (RekfCtl::Rec): ordered compaction by observation, and the DISTINCT matched landmarks in ascending order -- the rows of
// the sub-block of P that S = H P H^T + Q touches: rows 0, 1, 2, then (3 + 2 id, 4 + 2 id) per distinct landmark; two observations
// matched to ONE landmark (Q6) share its rows.  `rec` is global memory (the front end's last workgroup, for the k_mid behind the
// kernel boundary) or LDS (every mid workgroup for itself, when the front end runs inside k_mid's own grid).
// distinct_ranks: key = landmark id of a matched lane (0x7fffffff otherwise), rk = its rank among the keys (ties by lane), M matched
// lanes -> this lane's rank among the DISTINCT keys; lane r < M also learns the r-th smallest key and whether it repeats its predecessor
__device__ static inline int distinct_ranks(int key, int rk, int M, int lane, int &sorted, bool &dup, int &nu)
{
    sorted = __builtin_amdgcn_ds_permute(rk << 2, key);            // lane r: the r-th smallest key (r < M; unmatched lanes collide above M)
    const int prev = __shfl_up(sorted, 1, WAVE);
    dup = lane > 0 && lane < M && sorted == prev;
    const unsigned long long D = __ballot(dup);
    nu = M - __popcll(D);
    return rk - __popcll(D & ((2ull << (rk & 63)) - 1ull));       // repeats at sorted positions <= rk
}
__device__ static inline void compact_record(RekfCtl::Rec *rec, RekfCtl *ctl, int kind, int oidx, int lane, int K, int n, int n_max, int has_gps, bool raise_flag = true)
{
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const unsigned long long ms = __ballot(kind == 1);
    const unsigned long long mm = __ballot(kind == 0);
    const unsigned long long mn = __ballot(kind == 2);
    const int M = __popcll(ms), Mm = __popcll(mm);
    int N2 = __popcll(mn);
    const int room = (n_max - n) / 2;
    if (N2 > room) {                                               // capacity guard (ours)
        // (raise_flag = false: the SPECULATIVE front end -- its record is not the scan's until k_mid has proved it, and an observation it
        // calls new may match after all: k_mid raises the flag once the record stands)
        if (lane == 0 && raise_flag) atomicOr(&ctl->err, REKF_FLAG_CAPACITY);
        N2 = room;
    }
    int rk = 0;
    const int key = (kind == 1) ? oidx : 0x7fffffff;
#pragma unroll
    for (int q = 0; q < 32; ++q) {                                   // K <= 32 observations in a whole scan
        const int oq = __builtin_amdgcn_readlane(key, q);
        rk += (oq < key || (oq == key && q < lane)) ? 1 : 0;
    }
    int sorted, nu; bool dup;
    const int ur = distinct_ranks(key, rk, M, lane, sorted, dup, nu);
    {
        const unsigned long long D = __ballot(dup);                  // the sorted position `lane` holds a distinct landmark: its id, by distinct rank
        if (lane < M && lane < 32 && !dup) rec->uid[lane - __popcll(D & ((1ull << lane) - 1ull))] = sorted;
    }
    if (kind == 1) {
        const int p = __popcll(ms & lt);
        if (p < 32) { rec->pair_obs[p] = lane; rec->pair_id[p] = oidx; rec->pair_state[p] = 1; rec->urank[p] = ur; }
    } else if (kind == 0) {
        const int p = __popcll(mm & lt);
        if (M + p < 32) { rec->pair_obs[M + p] = lane; rec->pair_id[M + p] = oidx; rec->pair_state[M + p] = 0; }
    } else if (kind == 2) {
        const int p = __popcll(mn & lt);
        if (p < N2) rec->newid[p] = lane;
    }
    if (lane == 0) {
        const int MM = M + Mm;
        const int m = (MM > 0) ? 2 * MM + (has_gps ? 3 : 0) : 0;
        rec->cnt[0] = MM; rec->cnt[1] = m; rec->cnt[2] = (m + 15) & ~15; rec->cnt[3] = M; rec->cnt[4] = (has_gps && MM > 0) ? 1 : 0;
        rec->cnt[5] = N2; rec->cnt[6] = Mm; rec->cnt[7] = K;
        rec->nu = nu;
    }
}
// ReflectorMatch's MAP branch (cc:401-425) for one observation, by ONE wave: sqrt(e^T S e) against every pre-loaded point (S the stored
// covariance, not its inverse: quirk Q3), first minimum in index order, a match below 0.05.  The literal form: per-lane first minimum,
// then the wave-wide one (a NaN distance -- e^T S e < 0 for an indefinite S -- is treated as the reference's sort treats it here: never
// smaller than anything, but a lane's first candidate stands until a smaller one comes).
__device__ static inline void map_match_wave(const RekfDev &d, int mlane, float gx, float gy, int &kind, int &best_j)
{
#pragma clang fp contract(off)
    const int M_ = d.M_map;
    double best = 0; int bj = -1;
    for (int j = mlane; j < M_; j += 64) {
        const double *S = d.map_cov + 4 * (size_t)j;
        const float ex = d.map_xy[2 * j] - gx;
        const float ey = d.map_xy[2 * j + 1] - gy;
        const double dx = (double)ex, dy = (double)ey;
        const double t0 = dx * S[0] + dy * S[2];
        const double t1 = dx * S[1] + dy * S[3];
        const double dist = sqrt(t0 * dx + t1 * dy);
        if (bj < 0 || dist < best) { best = dist; bj = j; }
    }
    wave_argmin(best, bj);
    if (bj >= 0 && best < 0.05) { kind = 0; best_j = bj; }
}
__device__ static inline double readlane63_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// ... and for the SPECULATIVE front end: the same decision (map_lip >= 0: every S is symmetric positive semi-definite, no NaN can arise
// but from round-off -- which makes the observation unprovable: dm1 = NaN), with what the margin proof needs: the distance to the nearest
// map point (d1) and to the runner-up (d2)
__device__ static inline void map_match_wave_spec(const RekfDev &d, int mlane, float gx, float gy, int &kind, int &best_j, double &d1, double &d2)
{
#pragma clang fp contract(off)
    const int M_ = d.M_map;
    double b1 = 1e300, b2 = 1e300; int bj = -1; bool nan = false;
    for (int j = mlane; j < M_; j += 64) {
        const double *S = d.map_cov + 4 * (size_t)j;
        const float ex = d.map_xy[2 * j] - gx;
        const float ey = d.map_xy[2 * j + 1] - gy;
        const double dx = (double)ex, dy = (double)ey;
        const double t0 = dx * S[0] + dy * S[2];
        const double t1 = dx * S[1] + dy * S[3];
        const double dist = sqrt(t0 * dx + t1 * dy);
        nan = nan || !(dist == dist);
        b2 = vmin_f64(b2, vmax_f64(dist, b1));
        bj = (dist < b1) ? j : bj;
        b1 = vmin_f64(b1, dist);
    }
    double g1 = b1, g2 = b2; int gj = bj;
    argmin2_dpp_step<0x111, 0xf>(g1, gj, g2);
    argmin2_dpp_step<0x112, 0xf>(g1, gj, g2);
    argmin2_dpp_step<0x114, 0xf>(g1, gj, g2);
    argmin2_dpp_step<0x118, 0xf>(g1, gj, g2);
    argmin2_dpp_step<0x142, 0xa>(g1, gj, g2);
    argmin2_dpp_step<0x143, 0xc>(g1, gj, g2);
    g1 = readlane63_f64(g1); g2 = readlane63_f64(g2); gj = __builtin_amdgcn_readlane(gj, 63);
    d1 = g1; d2 = g2;
    if (__ballot(nan) != 0ull) { d1 = __longlong_as_double(0x7ff8000000000000ll); d2 = d1; return; }     // unprovable: k_mid re-matches it
    if (gj >= 0 && g1 < 0.05) { kind = 0; best_j = gj; }
}

// The front end as a ROLE of a workgroup of NT threads (a multiple of 256): k_front_mb below is nothing else; the fused kernel
// k_dd_front runs it in the workgroups behind its downdate workgroups.  corner_in_ctl: the pose block to predict from is
// RekfCtl::post_C9 (what k_mid evaluated for the previous scan) -- in k_dd_front the previous scan's downdate, which stores that block
// into P, is running beside this role; in k_front_mb it is read from P (set_state, reserve, a flushed host predict may have changed it).
// SPEC: the role runs for scan t + 1 inside scan t's launch (RekfCtl::spec): pose = the OLD mean moved by scan t's odometry (A.prev_*) and
// then its own; nothing of Predict is written (scan t + 1's k_mid evaluates its own); per observation the result AND the distances to the
// nearest reflector and the runner-up go to RekfCtl::spec[A.pred_slot], the compacted record to RekfCtl::rec[A.pred_slot].
template <int NT, bool SPEC = false>
__device__ __forceinline__ void front_role(const RekfDev &d, const RekfFrontArgs &A, const int b, const int nb, const bool corner_in_ctl)
{
    static_assert(NT >= 256 && NT % 256 == 0, "four waves match, lane 0 of wave 1 evaluates the motion model");
    __shared__ Motion mo;
    __shared__ double pose[5];
    __shared__ int s_last;                            // this workgroup matched the scan's last outstanding observation
    if (threadIdx.x == 0) s_last = 0;
#ifdef REKF_DEBUG_FRONT
    long long tqf[8]; int nqf = 0;
    const bool recf = b == 1 && threadIdx.x == 0;
    const long long t_entryf = clock64(), w_entryf = wall_clock64();
#define FMARK() do { __builtin_amdgcn_sched_barrier(0); if (recf && nqf < 8) tqf[nqf++] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FMARK()
#endif
    const int tid = threadIdx.x;
    RekfCtl *ctl = d.ctl;
    const double *__restrict__ P = d.P;
    const double *mu = d.mu;
    const size_t ld = (size_t)d.ld;
    int n = (d.n_known >= 0) ? d.n_known : ctl->n + (A.aug_pending == 1 ? 2 * ctl->n_new : 0);
    if (d.n_known < 0 && ((A.front_in_mid && A.aug_in_mid) || A.aug_pending == 2)) {
        // inside k_mid's grid beside the mid role whose workgroup 0 is appending the previous scan's reflectors (and moving ctl->n) right
        // now: the dimension comes from that scan's augmentation record, as the mid role takes it
        const RekfCtl::AugRec *ar = &ctl->augrec[(A.pred_slot ^ 1) & 1];
        n = ar->n_before + 2 * ar->n2;
    }
    const int L = (n - 3) / 2;
    const int K = A.K;
    // inside k_mid's grid with the motion model evaluated here (one launch per scan): what the mid role reads of this role's Predict goes
    // THROUGH to memory and is complete before this workgroup counts its observation (the mid role's wait for the count orders the rest)
    const bool wt_pred = !SPEC && A.front_in_mid != 0 && !A.host_pred;
    RekfCtl::Spec *const sp = &ctl->spec[A.pred_slot & 1];

    // operands that depend only on the old state go in flight first -- and this workgroup's first observation: a dynamically
    // indexed kernel argument is a scalar load of its own, issued where it is used (inside the match, it cost a memory round trip)
    float ob0x = 0.f, ob0y = 0.f;
    if (b < K) { ob0x = rekf_obs(A, 2 * b); ob0y = rekf_obs(A, 2 * b + 1); }
    // the first 1024 landmarks as float32 (cc:431) into LDS: these loads fly under the trig chain below, and the waves that match
    // then read LDS instead of waiting for HBM four times in a row
    __shared__ float s_lmx[1024], s_lmy[1024];
    {
        float lmx[1024 / NT], lmy[1024 / NT];
#pragma unroll
        for (int q = 0; q < 1024 / NT; ++q) {
            const int j = tid + NT * q;
            lmx[q] = 0.f; lmy[q] = 0.f;
            if (j < L) { lmx[q] = (float)mu[3 + 2 * j]; lmy[q] = (float)mu[4 + 2 * j]; }
        }
#pragma unroll
        for (int q = 0; q < 1024 / NT; ++q) { s_lmx[tid + NT * q] = lmx[q]; s_lmy[tid + NT * q] = lmy[q]; }
    }
    double C9[9];
    if (SPEC) {
        // two motion steps from the old pose: scan t's (A.prev_dt, A.prev_vt), then this scan's
        if (tid == 0) {
#pragma clang fp contract(off)
            const double th = mu[2] + A.prev_vt[2] * A.prev_dt + A.vt[2] * A.dt;
            double sn, cs;
            sincos(th, &sn, &cs);
            pose[2] = th; pose[3] = cs; pose[4] = sn;
        }
        if (tid == 64) {
#pragma clang fp contract(off)
            Motion m1;
            motion_terms_of(A.model, A.prev_dt, A.prev_vt[0], A.prev_vt[1], A.prev_vt[2], A.lin_cov, A.ang_cov, mu[2], m1);
            const double x1 = mu[0] + m1.d[0], y1 = mu[1] + m1.d[1], th1 = mu[2] + m1.d[2];
            motion_terms(A, th1, mo);
            pose[0] = x1 + mo.d[0]; pose[1] = y1 + mo.d[1];
        }
    } else if (A.host_pred) {
        // the host predicted (its pose mirror was current: rekf_api.hip): pose, cos / sin of the WRAPPED heading exactly as the
        // reference takes them (cc:181, :252-253), the composite (a, b) of every predict since the device last saw P and the
        // pose block come by value -- no motion model, no libm call in front of the match
        if (tid == 0) {
#pragma unroll
            for (int q = 0; q < 5; ++q) pose[q] = A.pre_pose[q];
            mo.a = A.pre_ab[0]; mo.b = A.pre_ab[1];
        }
    } else {
    if (tid == 0) {
        // cos / sin of the new heading on this lane; the motion terms, which have their own sincos, meanwhile on lane 0 of the
        // next wave.  DEVIATION (round-off level): the reference wraps the heading first, theta' = atan2(sin, cos)
        // (cc:181 / :205), and takes cos / sin of theta' wherever it needs them (cc:252-253, :390-391); here they are taken of
        // the unwrapped angle -- the same values up to the last place -- so that the match does not wait for a chain of three
        // libm calls (2.1 us) but for one; theta' itself (what is committed to the mean) is computed as written, after the barrier.
        // (Only on this path: when the host predicts, above, cos / sin are the reference's own.)
#pragma clang fp contract(off)
        if (b == 0) for (int q = 0; q < 9; ++q) C9[q] = corner_in_ctl ? ctl->post_C9[(A.pred_slot ^ 1) & 1][q] : rekf_plower(P, (int)ld, q % 3, q / 3);
        const double mu2 = mu[2];
        const double dth = A.vt[2] * A.dt;            // = mo.d[2] (delta_theta = w dt in both models; no FMA: same bits)
        double th = mu2 + dth, sn, cs;
        sincos(th, &sn, &cs);
        pose[2] = th; pose[3] = cs; pose[4] = sn;
        FMARK();                                      // 0: thread 0's sincos done
    }
    if (tid == 64) {
        const double mu0 = mu[0], mu1 = mu[1], mu2 = mu[2];
        motion_terms(A, mu2, mo);
        pose[0] = mu0 + mo.d[0]; pose[1] = mu1 + mo.d[1];
    }
    }
    __syncthreads();
    FMARK();                                          // 1: barrier passed

    // ---- Predict's covariance part (cc:178 / :202) is NOT applied here: (a, b) and the predicted pose block go to the control block,
    // k_mid and k_downdate2 apply them to what they read of P (RekfCtl::pred), and the scan's downdate commits them
    {
#pragma clang fp contract(off)
        if (SPEC) {
            if (b == 0 && tid == 0) { sp->pose[0] = pose[0]; sp->pose[1] = pose[1]; sp->pose[2] = pose[2]; sp->n = n; sp->scan = A.scan_id; }
        } else if (b == 0 && tid == 0) {
            if (A.host_pred) {
#pragma unroll
                for (int q = 0; q < 9; ++q) C9[q] = A.pre_C9[q];
            } else corner_predict(C9, 3, mo);
            RekfCtl::Pred *pr = &ctl->pred[A.pred_ix & 3];
            if (wt_pred) {
                store_wt(&pr->ab[0], mo.a); store_wt(&pr->ab[1], mo.b);
                for (int q = 0; q < 9; ++q) store_wt(&pr->C9[q], C9[q]);
                store_wt(&ctl->pose_pred[0], pose[0]); store_wt(&ctl->pose_pred[1], pose[1]);
                store_wt(&ctl->pose_pred[3], pose[3]); store_wt(&ctl->pose_pred[4], pose[4]);     // (drained in front of this lane's count, below)
            } else {
                pr->ab[0] = mo.a; pr->ab[1] = mo.b;
                for (int q = 0; q < 9; ++q) pr->C9[q] = C9[q];
                ctl->pose_pred[0] = pose[0]; ctl->pose_pred[1] = pose[1]; ctl->pose_pred[3] = pose[3]; ctl->pose_pred[4] = pose[4];
                if (A.host_pred) ctl->pose_pred[2] = pose[2];
            }
            if (!A.front_in_mid) ctl->pose_pending = 1;       // (inside k_mid's grid nobody reads it, and that kernel's workgroup 0 clears it beside us)
        }
        if (!SPEC && !A.host_pred && b == 0 && tid == NT - 64) {        // the wrapped heading (cc:181 / :205), on the last wave
            const double thw = atan2(pose[4], pose[3]);
            if (wt_pred) {
                store_wt(&ctl->pose_pred[2], thw);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (complete in front of the match's barrier, which lane 0 passes before it counts)
            } else ctl->pose_pred[2] = thw;
        }
    }

    FMARK();                                          // 2: covariance slice written
    // ---- ReflectorMatch (cc:370-455): whole observations per workgroup.  The state sweep of an observation is split over FOUR
    // waves (wave w takes the landmarks j0 + 64 w + lane: one candidate per lane and round, out of the LDS-staged means), each
    // reduces by DPP steps, the four partial (smallest, first index, second smallest) triples meet in LDS behind ONE block
    // barrier and wave 0 finishes (one wave alone needed 2.7 us per observation, the 1024-thread version with two-level
    // reductions 3 us).  The map match and the literal rescan stay on wave 0.
    const int M_ = d.M_map;
    __shared__ double s_part[2][4][2];
    __shared__ int s_partj[2][4];
    const int mwave = tid >> 6, mlane = tid & 63;
    int it = 0;
    for (int i = b; i < K; i += nb, ++it) {
        float gx = 0.f, gy = 0.f;
        const float obx = (it == 0) ? ob0x : rekf_obs(A, 2 * i), oby = (it == 0) ? ob0y : rekf_obs(A, 2 * i + 1);
        if (mwave < 4) obs_to_global(pose[0], pose[1], pose[3], pose[4], obx, oby, gx, gy);
        int kind = 2, best_j = -1;
        FMARK();                                      // m0: observation in the global frame
        double dm1v = 1e300, dm2v = 1e300;                           // SPEC: distance to the nearest map point / the runner-up
        if (mwave == 0 && M_ > 0) {                                // cc:401-425
            if constexpr (SPEC) map_match_wave_spec(d, mlane, gx, gy, kind, best_j, dm1v, dm2v);
            else map_match_wave(d, mlane, gx, gy, kind, best_j);
        }
        if (mwave < 4 && L > 0) {                                  // cc:426-451, this wave's quarter of the landmarks
#pragma clang fp contract(off)
            // smallest and second smallest SQUARED distance with the first index of the smallest: the sqrt is taken once, and
            // only if the two are within rounding of each other does the literal scan (sqrt per candidate) decide
            double b1 = 1e300, b2 = 1e300; int bj = -1;
            for (int j0 = 0; j0 < L; j0 += 256) {
                const int j = j0 + 64 * mwave + mlane, jc = j < L ? j : L - 1;
                float lx, ly;
                if (j0 < 1024) { lx = s_lmx[jc & 1023]; ly = s_lmy[jc & 1023]; }                  // staged above
                else { lx = (float)mu[3 + 2 * jc]; ly = (float)mu[4 + 2 * jc]; }        // cc:431
                const float ex = gx - lx, ey = gy - ly;                // cc:433
                const double dx = (double)ex, dy = (double)ey;
                const double d2 = (j < L) ? dx * dx + dy * dy : 1e300;
                b2 = vmin_f64(b2, vmax_f64(d2, b1));
                bj = (d2 < b1) ? j : bj;
                b1 = vmin_f64(b1, d2);
            }
            // wave-wide combine by six DPP steps (row shifts 1/2/4/8, row_bcast:15, row_bcast:31: lane 63 ends up with the
            // whole wave; a lane without a source sees the neutral triple)
            double g1 = b1, g2 = b2; int gj = bj;
            argmin2_dpp_step<0x111, 0xf>(g1, gj, g2);
            argmin2_dpp_step<0x112, 0xf>(g1, gj, g2);
            argmin2_dpp_step<0x114, 0xf>(g1, gj, g2);
            argmin2_dpp_step<0x118, 0xf>(g1, gj, g2);
            argmin2_dpp_step<0x142, 0xa>(g1, gj, g2);
            argmin2_dpp_step<0x143, 0xc>(g1, gj, g2);
            FMARK();                                  // m1: swept and reduced
            if (mlane == 63) { s_part[it & 1][mwave][0] = g1; s_part[it & 1][mwave][1] = g2; s_partj[it & 1][mwave] = gj; }
        }
        __syncthreads();        // (the buffer of round it is written again in round it + 2: the barrier of round it + 1 lies between)
        FMARK();                                      // m2: barrier passed
        if (mwave == 0) {
            double d1v = 1e300, d2v = 1e300;                           // SPEC: distance to the nearest reflector and to the runner-up
            if (kind == 2 && L > 0) {
#pragma clang fp contract(off)
                double g1 = s_part[it & 1][0][0], g2 = s_part[it & 1][0][1]; int gj = s_partj[it & 1][0];
#pragma unroll
                for (int w = 1; w < 4; ++w) argmin2_combine(g1, gj, g2, s_part[it & 1][w][0], s_partj[it & 1][w], s_part[it & 1][w][1]);
                double best = sqrt(g1);
                d1v = best; d2v = (g2 < 1e299) ? sqrt(g2) : 1e300;
                if (g2 <= g1 * 1.000000000000002) {                    // literal scan (uniform, rare)
                    d2v = d1v;                                         // (two candidates within rounding: nothing a margin could prove)
                    best = 0; gj = -1;
                    for (int j = mlane; j < L; j += 64) {
                        const float lx = (float)mu[3 + 2 * j], ly = (float)mu[4 + 2 * j];
                        const float ex = gx - lx, ey = gy - ly;
                        const double dx = (double)ex, dy = (double)ey;
                        const double dist = sqrt(dx * dx + dy * dy);   // cc:437
                        if (gj < 0 || dist < best) { best = dist; gj = j; }
                    }
                    wave_argmin(best, gj);
                }
                if (gj >= 0 && best < 0.6) { kind = 1; best_j = gj; }  // cc:446
            }
            if (mlane == 0) {
                // the result, coherently at agent scope (another workgroup -- on another XCD, behind another L2 -- may be the one that
                // compacts), then the count: whoever sees it reach the scan's target has every result in memory behind it
                if (SPEC) {
                    __hip_atomic_store(&sp->kind[i & 31], kind, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&sp->idx[i & 31], best_j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // (a map match: its own distances; else the state branch's, and the nearest map point's for the "no map match" half of the proof)
                    store_wt(&sp->d1[i & 31], kind == 0 ? dm1v : d1v); store_wt(&sp->d2[i & 31], kind == 0 ? dm2v : d2v);
                    store_wt(&sp->dm1[i & 31], dm1v);
                } else {
                    __hip_atomic_store(&ctl->obs_kind[i], kind, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&ctl->obs_idx[i], best_j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (!SPEC && A.compact_in_mid) continue;                // (the kernel boundary hands the results over: RekfFrontArgs::compact_in_mid)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (A.front_in_mid) (void)__hip_atomic_fetch_add(&ctl->front_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (nobody to elect: no returning round trip)
                else {
                    const unsigned old = __hip_atomic_fetch_add(&ctl->front_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (old + 1u == A.front_target) s_last = 1;
                }
            }
        }
    }
    FMARK();                                          // 3: match done
    // ---- the last workgroup to finish compacts the scan's results for the k_mid behind the kernel boundary (RekfCtl::rec).  Inside
    // k_mid's own grid (A.front_in_mid) nobody does: the mid workgroups wait for front_count and compact for themselves
    __syncthreads();
    if (A.compact_in_front && !A.front_in_mid && s_last && tid < 64) {
        const int lane = tid;
        const int *kp = SPEC ? &sp->kind[lane & 31] : &ctl->obs_kind[lane], *ip = SPEC ? &sp->idx[lane & 31] : &ctl->obs_idx[lane];
        const int kind = (lane < K) ? __hip_atomic_load(kp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
        const int oidx = (lane < K) ? __hip_atomic_load(ip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
        compact_record(&ctl->rec[A.pred_slot & 1], ctl, kind, oidx, lane, K, n, d.n_max, A.has_gps, !SPEC);
    }
#ifdef REKF_DEBUG_FRONT
    if (recf) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ctl->dbg[6] = clock64() - t_entryf; ctl->dbg[5] = wall_clock64() - w_entryf; ctl->dbg[7] = nqf;
        for (int i = 0; i < nqf; ++i) ctl->dbg[8 + i] = tqf[i] - t_entryf;
    }
#endif
}
__global__ __launch_bounds__(1024) void k_front_mb(RekfDev d, RekfFrontArgs A)
{
    // (1024 threads: one landmark each to stage; 256 do it in 1.1 us more.  A.corr: the stored P is a scan behind -- the pose block to
    // predict from is the pending scan's RekfCtl::post_C9)
    front_role<1024>(d, A, (int)blockIdx.x, (int)gridDim.x, A.corr != 0);
}

// ----------------------------------------------------------------------------
// H row pair p of the scan (cc:248-304, gps.cc:305-332): coefficients of rows 2p, 2p+1.
// ----------------------------------------------------------------------------
struct HPair {
    double a0[3], a1[3], b0[2], b1[2], dz0, dz1, q0, q1;
    int col;
};
__device__ static __forceinline__ HPair make_hpair(const RekfDev &d, const RekfFrontArgs &A, const double *pose,
                                   int local_id, int global_id, int is_state)
{
#pragma clang fp contract(off)
    HPair h;
    const double c = pose[3], s = pose[4];                      // cc:252-253
    const double z0 = (double)rekf_obs(A, 2 * local_id), z1 = (double)rekf_obs(A, 2 * local_id + 1);
    double lx, ly;
    const double *mul = d.mu_lin ? d.mu_lin : d.mu;          // the linearisation point (k_mid: later block steps of a wide scan)
    if (is_state) { lx = mul[3 + 2 * global_id]; ly = mul[4 + 2 * global_id]; }
    else { lx = (double)d.map_xy[2 * global_id]; ly = (double)d.map_xy[2 * global_id + 1]; }
    const double dx = lx - pose[0], dy = ly - pose[1];         // cc:267-268
    const double zh0 = dx * c + dy * s, zh1 = -dx * s + dy * c; // cc:269-270
    h.a0[0] = -c; h.a0[1] = -s; h.a0[2] = -dx * s + dy * c;    // A_i cc:272-273
    h.a1[0] = s;  h.a1[1] = -c; h.a1[2] = -dx * c - dy * s;
    h.b0[0] = c;  h.b0[1] = s;                                 // B cc:255 (state rows only, cc:275)
    h.b1[0] = -s; h.b1[1] = c;
    h.col = is_state ? 3 + 2 * global_id : -1;
    h.dz0 = z0 - zh0; h.dz1 = z1 - zh1;
    h.q0 = A.obs_cov; h.q1 = A.obs_cov;                         // cc:276 / :302
    return h;
}

// ----------------------------------------------------------------------------
// The m x m solve (inside k_mid): S^-1 by blocked Gauss-Jordan.
//
// The inverse is a chain of m sequential pivots, so it is latency-bound; the design minimises what sits on that chain:
//   * S lives in registers as 16x16 blocks in the v_mfma_f64_16x16x4_f64 C/D layout (lane (g,c) = (lane>>4, lane&15),
//     register r <-> element (g+4r, c)); wave w owns block COLUMN w (one wave per SIMD for m <= 64).
//   * Block step K: wave K publishes its column blocks S(i,K) to LDS, inverts the diagonal block entirely in-wave (2x2
//     pivots, 8 steps, pivot rows / columns exchanged through a wave-private LDS scratch, no barrier) and publishes
//     D^-1: ONE LDS barrier per block step.  Every other wave j then forms its block of the pivot row,
//     R = D^-1 S(K,j), and applies S(i,j) -= S(i,K) R with MFMA; wave K+1 does block (K+1,K+1) first and goes straight
//     into the next leaf, so the chain per step is leaf + 2 block products -- the other products run in the shadow of
//     the leaf.  A block in C layout is directly an MFMA B operand; the A operand is the C layout of the transposed
//     block, read with transposed addressing from the row-major 16x17 patch the owner published.
// ----------------------------------------------------------------------------
__device__ static inline void lds_barrier()
{
    // LDS-only workgroup barrier: does not wait for outstanding global memory traffic (vmcnt)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// In-wave Gauss-Jordan inverse of one 16x16 block in C layout, 2x2 pivots (the rows come in (x,y)
// pairs).  Step with pivot block K = {k,k+1}, D = A(K,K): with the pivot columns replaced by unit
// vectors, R = D^-1 A(K,:), A(i,:) -= A(i,K) R for i not in K, A(K,:) = R.  Pivot rows and columns
// are exchanged through a 64-double LDS scratch private to the wave (LDS operations of one wave
// execute in order: no barrier, 5 writes + 7 16-byte reads per step).  Returns true when a pivot
// block is not positive definite (S must be: it is H P H^T + Q).  side(kk) is called once per step with
// independent work of the caller (MFMAs, LDS publishes) that is issued under the step's latency.
#define REKF_LEAF_SCRATCH 64
struct NoSideWork { __device__ void operator()(int) const {} };
template <class Side>
__device__ static inline bool leaf_inverse16(v4d &a, int g, int c, double *lp, Side &&side)
{
    bool bad = false;
    double *rowbuf = lp;            // [col c][row k / k+1]
    double *colbuf = lp + 32;       // [g][r][col k / k+1]
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int k = 2 * kk, rk = k >> 2, gk = k & 3;      // rows k, k+1 live in lane groups gk, gk+1, register rk
        if (g == gk || g == gk + 1) rowbuf[2 * c + (g - gk)] = a[rk];
        if (c == k || c == k + 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) colbuf[(4 * g + r) * 2 + (c - k)] = a[r];
        }
        __builtin_amdgcn_wave_barrier();
        const v2d dA = *(const v2d *)(rowbuf + 2 * k), dB = *(const v2d *)(rowbuf + 2 * k + 2);
        const v2d xx = *(const v2d *)(rowbuf + 2 * c);
        v2d ff[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ff[r] = *(const v2d *)(colbuf + (4 * g + r) * 2);
        __builtin_amdgcn_wave_barrier();
        side(kk);                                           // independent work that fills this step's latency
        const double d00 = dA.x, d10 = dA.y, d01 = dB.x, d11 = dB.y;
        const double det = d00 * d11 - d01 * d10;
        if (!(det > 0.0) || !(d00 > 0.0)) bad = true;
        double q = __builtin_amdgcn_rcp(det);               // 1/det: hardware reciprocal + two Newton steps
        q = fma(q, fma(-det, q, 1.0), q);
        q = fma(q, fma(-det, q, 1.0), q);
        const double i00 = d11 * q, i01 = -d01 * q, i10 = -d10 * q, i11 = d00 * q;
        const bool pc0 = c == k, pc1 = c == k + 1;          // pivot columns act as unit vectors
        double x0 = xx.x, x1 = xx.y;
        if (pc0) { x0 = 1.0; x1 = 0.0; }
        if (pc1) { x0 = 0.0; x1 = 1.0; }
        const double R0 = i00 * x0 + i01 * x1, R1 = i10 * x0 + i11 * x1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double base = (pc0 || pc1) ? 0.0 : a[r];
            double v = fma(-ff[r].y, R1, fma(-ff[r].x, R0, base));
            if (r == rk) v = (g == gk) ? R0 : ((g == gk + 1) ? R1 : v);
            a[r] = v;
        }
    }
    return bad;
}

// 16x17 row-major LDS patch <-> C layout; read_patch_T gives the C layout of the TRANSPOSED block,
// i.e. the block as an MFMA A operand (register q = k-slice q).
#define REKF_PATCH (16 * 17)
__device__ static inline void write_patch(double *patch, const v4d &a, int g, int c)
{
#pragma unroll
    for (int r = 0; r < 4; ++r) patch[(g + 4 * r) * 17 + c] = a[r];
}
__device__ static inline v4d read_patch_T(const double *patch, int g, int c)
{
    v4d t;
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = patch[c * 17 + g + 4 * q];
    return t;
}

// acc + A B for 16x16 blocks: At = C layout of A^T (register q = k-slice q), B in C layout.
__device__ static inline v4d block_mma(const v4d &At, const v4d &B, v4d acc)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(At[q], B[q], acc, 0, 0, 0);
    return acc;
}

#ifdef REKF_DEBUG_TIMING
__device__ static inline long long pinned_clock()
{
    __builtin_amdgcn_sched_barrier(0);
    const long long t = clock64();
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#endif
// same product as two independent accumulation chains (a dependent MFMA costs ~100 cycles, an
// independent one 64): for the two products that sit on the pivot chain
__device__ static inline v4d block_mma2(const v4d &At, const v4d &B, v4d acc)
{
    v4d acc1 = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(At[0], B[0], acc, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(At[2], B[2], acc1, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(At[1], B[1], acc, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(At[3], B[3], acc1, 0, 0, 0);
    return acc + acc1;
}

// Blocked Gauss-Jordan inverse of the (16 nbr) x (16 nbr) matrix whose block column w this wave holds in S[] (C layout),
// in place; the scheme is described above k_solve.  EVERY wave of the workgroup must call it (one s_barrier per block
// step); waves with w >= nbr only keep the barrier count.  Returns true when a pivot block was not positive definite.
struct NoMark { __device__ void operator()() const {} };
template <int NBR, class Idle, class Mark = NoMark>
__device__ static inline bool gj_invert_blocks(v4d (&S)[NBR], int nbr, int w, int g, int c, double (*s_col)[NBR + 1][REKF_PATCH], double *lp, Idle &&idle, Mark &&mark = Mark())
{
    // idle(slot): the waves 4.. (no block column; they only keep the barrier count) are called back before the first barrier (slot 0) and
    // behind each one (slot K + 1) with work of the caller that must not sit on the pivot chain
    bool bad = false;
    const v4d zero4 = {0, 0, 0, 0};
    if (w >= 4) idle(0);
    // wave 0 opens the chain: publish column 0, invert S(0,0)
    if (w == 0) {
#pragma unroll
        for (int bi = 1; bi < NBR; ++bi)
            if (bi < nbr) write_patch(s_col[0][bi], S[bi], g, c);
        bad |= leaf_inverse16(S[0], g, c, lp, NoSideWork());
        write_patch(s_col[0][NBR], S[0], g, c);
    }
#pragma unroll
    for (int K = 0; K < NBR; ++K) {
        if (K >= nbr) break;
        double (*col)[REKF_PATCH] = s_col[K & 1];
        mark();                                       // (debug builds: wave 0 reaches the barrier ...)
        lds_barrier();                                // column K and D^-1 of step K are published
        mark();                                       // (... and leaves it)
        if (w >= nbr) { if (w >= 4) idle(K + 1); continue; }      // a wave without a block column only keeps the barrier count
        if (w == K) {
            // S(i,K) <- -S(i,K) D^-1 ; S(K,K) = D^-1 is already in place
#pragma unroll
            for (int bi = 0; bi < NBR; ++bi)
                if (bi != K && bi < nbr) S[bi] = block_mma(-read_patch_T(col[bi], g, c), S[K], zero4);
        } else {
            // pivot-row block R = D^-1 S(K,w), then S(i,w) -= S(i,K) R, block (K+1, .) first
            const v4d R = block_mma2(read_patch_T(col[NBR], g, c), S[K], zero4);
            S[K] = R;
            if (K + 1 < NBR && K + 1 < nbr)
                S[K + 1] = block_mma2(-read_patch_T(col[K + 1], g, c), R, S[K + 1]);
            if (K + 1 < NBR && w == K + 1) {
                // Next pivot wave: invert the diagonal block now.  Its other NBR-2 products and the
                // publication of column K+1 (other buffer) are issued from inside the leaf, under
                // the latency of its pivot steps, so they leave the pivot chain.
                constexpr int NP = (NBR > 2) ? NBR - 2 : 0;            // pending products, blocks bi not in {K, K+1}
                constexpr int PH = (NBR <= 4) ? 4 : 6;                 // leaf steps that carry MFMAs; the rest carry publishes
                constexpr int PER = (4 * NP + PH - 1) / PH, PP = (NBR - 1 + 7 - PH) / (8 - PH);
                double (*ncol)[REKF_PATCH] = s_col[(K + 1) & 1];
                v4d At[NP > 0 ? NP : 1];
#pragma unroll
                for (int idx = 0; idx < NP; ++idx) {
                    const int bi = (idx >= K) ? idx + 2 : idx;
                    At[idx] = (bi < nbr) ? -read_patch_T(col[bi], g, c) : zero4;
                }
                auto side = [&](int kk) {
                    if (kk < PH) {
#pragma unroll
                        for (int t = 0; t < PER; ++t) {
                            const int q = kk * PER + t;                // consecutive MFMAs belong to different products
                            if (q < 4 * NP) {
                                const int idx = q % (NP > 0 ? NP : 1), slice = q / (NP > 0 ? NP : 1);
                                const int bi = (idx >= K) ? idx + 2 : idx;
                                if (bi < nbr)
                                    S[bi] = __builtin_amdgcn_mfma_f64_16x16x4f64(At[idx][slice], R[slice], S[bi], 0, 0, 0);
                            }
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < PP; ++t) {
                            const int pidx = (kk - PH) * PP + t;
                            if (pidx < NBR - 1) {
                                const int bi = (pidx >= K + 1) ? pidx + 1 : pidx;
                                if (bi < nbr) write_patch(ncol[bi], S[bi], g, c);
                            }
                        }
                    }
                };
                bad |= leaf_inverse16(S[K + 1], g, c, lp, side);
                write_patch(ncol[NBR], S[K + 1], g, c);
            } else {
#pragma unroll
                for (int bi = 0; bi < NBR; ++bi)
                    if (bi != K && bi != K + 1 && bi < nbr) S[bi] = block_mma(-read_patch_T(col[bi], g, c), R, S[bi]);
            }
        }
    }
    return bad;
}

// ----------------------------------------------------------------------------
// k_compact_wide: the ordered compaction of the per-observation match results of a WIDE scan (more than
// REKF_MAX_OBS_DEV observations; up to REKF_MAX_OBS_WIDE) into the ReflectorMatchResult lists of the control block
// (cc:397-453: state matches, map matches and new observations, each in increasing observation order).  k_mid then
// takes the matched pairs a block at a time.  One workgroup, one thread per observation.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(REKF_MAX_OBS_WIDE) void k_compact_wide(RekfDev d, RekfFrontArgs A)
{
    __shared__ int s_w[3][REKF_MAX_OBS_WIDE / 64];
    RekfCtl *ctl = d.ctl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = A.K;
    const int n = (d.n_known >= 0) ? d.n_known : ctl->n;
    const int kind = (tid < K) ? ctl->obs_kind[tid] : -1;
    const int oidx = (tid < K) ? ctl->obs_idx[tid] : -1;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const unsigned long long ms = __ballot(kind == 1), mm = __ballot(kind == 0), mn = __ballot(kind == 2);
    if (lane == 0) { s_w[0][wave] = __popcll(ms); s_w[1][wave] = __popcll(mm); s_w[2][wave] = __popcll(mn); }
    __syncthreads();
    int b0 = 0, b1 = 0, b2 = 0, M = 0, Mm = 0, N2 = 0;
#pragma unroll
    for (int w = 0; w < REKF_MAX_OBS_WIDE / 64; ++w) {
        if (w < wave) { b0 += s_w[0][w]; b1 += s_w[1][w]; b2 += s_w[2][w]; }
        M += s_w[0][w]; Mm += s_w[1][w]; N2 += s_w[2][w];
    }
    const int room = (d.n_max - n) / 2;
    if (N2 > room) {                                               // capacity guard (ours)
        if (tid == 0) atomicOr(&ctl->err, REKF_FLAG_CAPACITY);
        N2 = room;
    }
    if (kind == 1) { const int p = b0 + __popcll(ms & lt); ctl->state_pairs[2 * p] = tid; ctl->state_pairs[2 * p + 1] = oidx; }
    else if (kind == 0) { const int p = b1 + __popcll(mm & lt); ctl->map_pairs[2 * p] = tid; ctl->map_pairs[2 * p + 1] = oidx; }
    else if (kind == 2) { const int p = b2 + __popcll(mn & lt); if (p < N2) ctl->new_ids[p] = tid; }
    if (tid == 0) {
        const int MM = M + Mm;
        const int m = (MM > 0) ? 2 * MM + (A.has_gps ? 3 : 0) : 0;
        ctl->K = K; ctl->n_state = M; ctl->n_map = Mm; ctl->n_new = N2;
        ctl->m = m; ctl->m_pad = (m + 15) & ~15;
    }
}

// ----------------------------------------------------------------------------
// Landmark augmentation (cc:311-364), the covariance rows of a scan's new reflectors: by k_augment (one workgroup, behind the
// scan's downdate) or by workgroup 0 of the NEXT scan's k_mid (RekfCtl::augrec).
// ----------------------------------------------------------------------------
// The rows: every thread of the calling workgroup (nt of them).  Gp [N2][6], Sxi [9], RQR [4] are the caller's LDS; obs(k, rx, ry)
// yields the k-th new reflector's observation.  Ends on a workgroup barrier; the caller commits n.
template <class Obs>
__device__ static inline void augment_rows(const RekfDev &d, int n, int N2, double obs_cov, double (*Gp)[6], double *Sxi, double *RQR, int nt, Obs &&obs)
{
    const int tid = threadIdx.x;
    const size_t ld = (size_t)d.ld;
    double *P = d.P;
    {
#pragma clang fp contract(off)
        const double th = d.mu[2];
        const double s = sin(th), c = cos(th);                      // cc:323-324
        if (tid < N2) {
            // (the means of the new reflectors, cc:341-342, are already there: k_mid's workgroup 0 writes them behind its pose commit)
            float fx, fy;
            obs(tid, fx, fy);                                       // cc:338
            const double rx = (double)fx, ry = (double)fy;
            Gp[tid][0] = 1.; Gp[tid][1] = 0.; Gp[tid][2] = -rx * s - ry * c;   // cc:347
            Gp[tid][3] = 0.; Gp[tid][4] = 1.; Gp[tid][5] = rx * c - ry * s;
        }
        if (tid == 0) {
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) Sxi[i * 3 + j] = rekf_plower(P, (int)ld, i, j);   // cc:322
            const double q = obs_cov;                               // Gz Qt Gz^T, Gz = R(theta) (cc:326,354)
            RQR[0] = c * q * c + (-s) * q * (-s); RQR[1] = c * q * s + (-s) * q * c;
            RQR[2] = s * q * c + c * q * (-s);    RQR[3] = s * q * s + c * q * c;
        }
    }
    __syncthreads();
    // sigma_mx = G_fx * sigma (cc:355-357): rows n+2a+rr, all old columns (below the diagonal: the only copy that is stored);
    // sigma(0..2, col) is read as sigma(col, 0..2) -- the coalesced columns
    for (int e = tid; e < n * N2; e += nt) {
#pragma clang fp contract(off)
        const int a = e / n, col = e - a * n;
        const double q0 = rekf_plower(P, (int)ld, col, 0), q1 = rekf_plower(P, (int)ld, col, 1), q2 = rekf_plower(P, (int)ld, col, 2);
        for (int rr = 0; rr < 2; ++rr) {
            double acc = 0;
            acc += Gp[a][rr * 3 + 0] * q0;
            acc += Gp[a][rr * 3 + 1] * q1;
            acc += Gp[a][rr * 3 + 2] * q2;
            P[(size_t)(n + 2 * a + rr) + (size_t)col * ld] = acc;
        }
    }
    // sigma_mm (cc:354,358): every (a,b) block, a != b included, gets + R Qt R^T
    for (int e = tid; e < N2 * N2; e += nt) {
#pragma clang fp contract(off)
        const int a = e / N2, b = e - a * N2;
        for (int rr = 0; rr < 2; ++rr)
            for (int cc = 0; cc < 2; ++cc) {
                double acc = 0;
                for (int k = 0; k < 3; ++k) {
                    double t = 0;
                    for (int l = 0; l < 3; ++l) t += Gp[a][rr * 3 + l] * Sxi[l * 3 + k];
                    acc += t * Gp[b][cc * 3 + k];
                }
                const size_t gi = (size_t)(n + 2 * a + rr), gj = (size_t)(n + 2 * b + cc);
                if (gi < gj) continue;                              // the lower triangle is what is stored
                const double v = acc + RQR[rr * 2 + cc];
                P[gi + gj * ld] = v;
            }
    }
    __syncthreads();
}
// ReflectorMatch's state branch (cc:426-451) for ONE observation, by a whole workgroup of >= 256 threads (waves 0..3 sweep the landmarks,
// every thread must call; two workgroup barriers): the exact re-match of an observation whose speculative result k_mid could not prove
// (RekfCtl::spec).  Same arithmetic as front_role's sweep -- smallest and second smallest squared distance, the literal scan on a tie.
__device__ static inline void rematch_obs(const RekfDev &d, const double *mu, int L, float gx, float gy, int &kind, int &best_j)
{
    __shared__ double s_rp[4][2];
    __shared__ int s_rj[4], s_rk[2];
    const int tid = threadIdx.x, mwave = tid >> 6, mlane = tid & 63;
    int kind_m = 2, idx_m = -1;                          // the map branch comes first (cc:401-425): wave 0, before it finishes the state branch
    if (mwave == 0 && d.M_map > 0) map_match_wave(d, mlane, gx, gy, kind_m, idx_m);
    if (mwave < 4) {
#pragma clang fp contract(off)
        double b1 = 1e300, b2 = 1e300; int bj = -1;
        for (int j0 = 0; j0 < L; j0 += 256) {
            const int j = j0 + 64 * mwave + mlane, jc = j < L ? j : L - 1;
            const float lx = (float)mu[3 + 2 * jc], ly = (float)mu[4 + 2 * jc];        // cc:431
            const float ex = gx - lx, ey = gy - ly;                                   // cc:433
            const double dx = (double)ex, dy = (double)ey;
            const double d2 = (j < L) ? dx * dx + dy * dy : 1e300;
            b2 = vmin_f64(b2, vmax_f64(d2, b1));
            bj = (d2 < b1) ? j : bj;
            b1 = vmin_f64(b1, d2);
        }
        double g1 = b1, g2 = b2; int gj = bj;
        argmin2_dpp_step<0x111, 0xf>(g1, gj, g2);
        argmin2_dpp_step<0x112, 0xf>(g1, gj, g2);
        argmin2_dpp_step<0x114, 0xf>(g1, gj, g2);
        argmin2_dpp_step<0x118, 0xf>(g1, gj, g2);
        argmin2_dpp_step<0x142, 0xa>(g1, gj, g2);
        argmin2_dpp_step<0x143, 0xc>(g1, gj, g2);
        if (mlane == 63) { s_rp[mwave][0] = g1; s_rp[mwave][1] = g2; s_rj[mwave] = gj; }
    }
    __syncthreads();
    if (mwave == 0) {
#pragma clang fp contract(off)
        int kd = kind_m, bjj = idx_m;
        if (kind_m == 2 && L > 0) {
            double g1 = s_rp[0][0], g2 = s_rp[0][1]; int gj = s_rj[0];
#pragma unroll
            for (int w = 1; w < 4; ++w) argmin2_combine(g1, gj, g2, s_rp[w][0], s_rj[w], s_rp[w][1]);
            double best = sqrt(g1);
            if (g2 <= g1 * 1.000000000000002) {                    // literal scan (uniform, rare)
                best = 0; gj = -1;
                for (int j = mlane; j < L; j += 64) {
                    const float lx = (float)mu[3 + 2 * j], ly = (float)mu[4 + 2 * j];
                    const float ex = gx - lx, ey = gy - ly;
                    const double dx = (double)ex, dy = (double)ey;
                    const double dist = sqrt(dx * dx + dy * dy);   // cc:437
                    if (gj < 0 || dist < best) { best = dist; gj = j; }
                }
                wave_argmin(best, gj);
            }
            if (gj >= 0 && best < 0.6) { kd = 1; bjj = gj; }      // cc:446
        }
        if (mlane == 0) { s_rk[0] = kd; s_rk[1] = bjj; }
    }
    __syncthreads();
    kind = s_rk[0]; best_j = s_rk[1];
}

// ----------------------------------------------------------------------------
// k_mid<NBR, MODE>: gather + solve + gain in ONE launch, for at most 16 NBR innovation rows per pass (NBR = 2: up to 16
// matched observations, NBR = 4: up to 32 -- every BASELINE.json configuration; scans with more run several passes,
// see "block step" below).
//
// Round 1 ran k_gather -> k_solve -> k_gain: 24.4 us of kernels at C3 plus two kernel boundaries, of which the
// single-workgroup solve alone was 12.9 us while 255 CUs idled.  The inverse is a LATENCY chain, not work: here
// every workgroup owns 16 state rows and redoes it for itself -- the 67 x 67 sub-block of P that S = H P H^T + Q
// touches is L2 hits for all but the first -- and then needs neither a launch boundary nor a trip through memory for S^-1:
//   A  the scan's match record (ordered pairs, the distinct matched landmarks in ascending order), H rows packed in LDS;
//   C  the sub-block P(R, R), R = {0, 1, 2, rows of the distinct matched landmarks}, and this workgroup's rows P(own, R), gathered
//      IN THE MFMA C LAYOUT (round 5): the stored covariance is ONE SCAN BEHIND the filter (ekf_dev.h) -- a scan's downdate stays
//      pending as its Kn / HPt panels and runs as a ROLE of the next scan's launch, beside this one, from one P buffer into the
//      other -- so every gathered element takes, in the downdate's own arithmetic (the same v_mfma_f64_16x16x4 chain over k, then
//      P + sum), the pending Predict and the pending rank-m correction sum_k HPt(min, k) Kn(max, k): the bits the downdate stores;
//   D  W = P H^T (rows of S, own rows) and (H P)^T (own columns, straight to HBM for the downdate) out of LDS;
//   E  S = H W + Q in the MFMA C layout, blocked Gauss-Jordan (gj_invert_blocks), S^-1 -> LDS;
//   F  K(16 rows) = W S^-1 by MFMA out of LDS; Kn = -K -> HBM; mu += K (z - zhat) -- the reference's own
//      association K_t * (z - z_hat), cc:306 -- pose commit, theta wrap.
// W itself is never written.  All workgroups compute bit-identical S^-1 (same code, same inputs, no atomics).
// ----------------------------------------------------------------------------
typedef double v2du __attribute__((ext_vector_type(2), aligned(8)));     // a row pair that starts on an odd row: 8-byte aligned
#define MID_ROWS 16
// All of k_mid's LDS is ONE dynamic arena (the downdate role and the mid role of a launch lay their own structures over it)
template <int NBR> struct MidLds {
    static constexpr int MP = 16 * NBR;           // most innovation rows (padded) this instance takes
    static constexpr int NPAIR = MP / 2;
    static constexpr int NUS = 3 + 2 * NPAIR;     // rows of the sub-block: 0, 1, 2, then two per DISTINCT matched landmark, ascending
    static constexpr int NUSP = NUS + 1;          // its row stride in LDS (even: 16-byte pairs start on the odd columns 3 + 2 u, read as 8-byte-aligned vectors)
    static constexpr int NBU = (NUS + 15) / 16;   // 16 x 16 blocks per side
    static constexpr int LDS_S = MP + 16;         // row stride of S^-1 in LDS: = 16 mod 32 doubles, so the 4 k-rows of an MFMA operand read hit disjoint banks
    static constexpr int SS = 16 * NBU + 4;       // row stride of the staged panel rows ([k][sub-block row]; the MFMA lanes of a padded block read past NUS)
    struct Work {                                 // phases D .. F
        double s_col[2][NBR + 1][REKF_PATCH];
        double s_leaf[4][REKF_LEAF_SCRATCH];
        alignas(16) double s_wc0[3][MP];          // rows 0..2 of W
        alignas(16) double s_wcp[NPAIR][MP][2];   // state pair p: its two landmark rows of W, interleaved per column
        alignas(16) double s_wown[MP][MID_ROWS];  // this workgroup's rows of W, [column r][row]
        double s_dmu[4][MID_ROWS];
        double s_dc[4][3][MID_ROWS];              // workgroup 0: partial sums of (K H P)(i, jc), jc = 0..2, per wave of phase F
    };
    union U {
        Work w;
        double s_stage[2][REKF_PANEL_COLS][SS];   // phase C: rows R of the pending HPt [0] / Kn [1] panels, [k][sub-block row]
    } u;
    alignas(16) double s_coef[8 * MP];            // H row r in 64 bytes
    // the sub-block as gathered (+ pending Predict, + pending correction), both triangles; dead once W is formed: S^-1 (phase E) takes its place
    alignas(16) double s_big[((NUS + 1) * NUSP > MP * LDS_S) ? (NUS + 1) * NUSP : MP * LDS_S];
    alignas(16) double s_pw[16 * NBU][MID_ROWS];  // P(own rows, sub-block row s), [s][own row]
    RekfCtl::Rec s_rec;                           // the scan's match record
    int s_pcol[NPAIR];                            // pair -> its landmark's first sub-block row (3 + 2 urank), or -1 (map pair)
    int s_upair[NPAIR];                           // distinct landmark (by rank) -> a state pair that observes it
    int s_ownsub[MID_ROWS];                       // own row -> its sub-block row, or -1
    int s_fk[32], s_fi[32];                       // ... and the exact results of the observations whose speculative match could not be proved
    double s_np[3];                               // the committed pose, for the new reflectors' means
    double s_pred[12];                            // this scan's Predict: ab[0], ab[1], C9[0..8]
    double s_cpred[12];                           // the PENDING scan's (a, b) [0..1] and its pose block after the update [2..10]
};
constexpr int REKF_DD_LDS_BYTES = 4 * 64 * 64 * (int)sizeof(double);      // the downdate role's four panels (KC = 64)

// MODE picks what else the launch hosts (separate instantiations: the steady state of a full filter, MODE 0, carries none of it):
//   0  nothing;  1  a filter that can still grow: workgroup 0 leaves the scan's augmentation record (RekfCtl::augrec) and, with
//   A.aug_in_mid, first appends the PREVIOUS scan's new reflectors;  2  the scan's front end runs as the first A.front_in_mid
//   workgroups of this grid; also leaves the augmentation record;  3  like 0, and every workgroup may match the scan itself through the
//   match grid (A.grid_match; MODE 1 can, too).
// ONE LAUNCH PER SCAN (round 5, MODE 0 and 2): with A.dd_in_mid the workgroups from A.dd_first on are the PREVIOUS scan's downdate
// (dd_body<64> on four of their eight waves, tiles from a queue, from dp.P into dp.P_out); the mid role reads dp.P = d.P and takes
// the pending correction from dp's panels (A.corr); nobody waits for anybody inside the launch except the mid role for an in-grid
// front end (MODE 2: exclusive handles only, rekf_api.hip).
template <int KC, bool QUEUE> __device__ __forceinline__ void dd_body(const RekfDev &d, double *dd_smem, int wg, int nwg, unsigned *queue, bool pub_wg);
template <int NBR, int MODE>
__global__ __launch_bounds__(512) void k_mid(RekfCtl *ctl_first, int h0, int h1, int h2, int h3, int h4, int h5, RekfDev d_arg, RekfFrontArgs A_arg, RekfDev dp_arg, RekfFrontArgs An_arg)
{
    // h0 .. h5: the handful of launch-packet fields the kernel's FIRST instructions need (which role a workgroup has; which parity slots of
    // the control block the mid role's first loads read), packed by mid_head() into six leading scalar arguments: with
    // -amdgpu-kernarg-preload-count they arrive in SGPRs with the wave, so those loads issue without a trip to the kernel-argument segment
    const int hd_pred_slot = h0 & 1, hd_pred_ix = (h0 >> 1) & 3, hd_corr = (h0 >> 3) & 1, hd_corr_post = (h0 >> 4) & 1, hd_corr_pred_ix = (h0 >> 5) & 3,
              hd_spec = (h0 >> 7) & 1, hd_cpred = (h0 >> 8) & 1, hd_dd_par = (h0 >> 9) & 1;
    const int hd_K = h1;
    const unsigned hd_scan_id = (unsigned)h2, hd_corr_scan = (unsigned)h3;
    const int hd_dd_first = h4 & 0xfff, hd_n_mid = (h4 >> 12) & 0xfff, hd_spec_front = (h4 >> 24) & 0xff;
    const int hd_dd_in_mid = h5 & 0xfff, hd_front_in_mid = (h5 >> 12) & 0xff;
    // The four argument structs are read THROUGH the kernel-argument segment, where a role needs a field: named as parameters, every field
    // any role uses is fetched into SGPRs (and spilled) by every wave at the kernel's entry -- in front of the mid role's first loads, the
    // head of the update's critical chain
    static_assert(sizeof(RekfDev) % 8 == 0 && sizeof(RekfFrontArgs) % 8 == 0, "kernel-argument offsets below");
    const char *const kargs = (const char *)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int KOFF = 8 + 6 * 4;               // ctl pointer + six ints (= 32: the structs keep their 8-byte alignment)
    const RekfDev &d = *(const RekfDev *)(kargs + KOFF);
    const RekfFrontArgs &A = *(const RekfFrontArgs *)(kargs + KOFF + sizeof(RekfDev));
    const RekfDev &dp = *(const RekfDev *)(kargs + KOFF + sizeof(RekfDev) + sizeof(RekfFrontArgs));
    const RekfFrontArgs &An = *(const RekfFrontArgs *)(kargs + KOFF + 2 * sizeof(RekfDev) + sizeof(RekfFrontArgs));
    (void)d_arg; (void)A_arg; (void)dp_arg; (void)An_arg;
    // (MODE 3 = MODE 0 + the in-kernel grid match: an instantiation of its own, so that the steady-state kernel of the speculation pipeline
    // -- which never matches anything itself -- keeps its register budget)
    constexpr bool FRONT = MODE == 2, AUGR = MODE == 1, AUGW = MODE == 1 || MODE == 2, DDROLE = MODE != 1, GM = MODE == 1 || MODE == 3;
    // ctl_first = d.ctl, as a leading pointer argument of its own: built with -mllvm -amdgpu-kernarg-preload-count the wave starts with it
    // in SGPRs, and the kernel's first loads (the match results) do not wait for the kernel-argument fetch
    using LT = MidLds<NBR>;
    constexpr int MP = LT::MP, NPAIR = LT::NPAIR, NUSP = LT::NUSP, NBU = LT::NBU, LDS_S = LT::LDS_S;
    extern __shared__ __attribute__((aligned(16))) double k_mid_arena[];
    LT &L = *reinterpret_cast<LT *>(k_mid_arena);
    auto &s_col = L.u.w.s_col; auto &s_leaf = L.u.w.s_leaf; auto &s_coef = L.s_coef; auto &s_wc0 = L.u.w.s_wc0; auto &s_wcp = L.u.w.s_wcp;
    auto &s_wown = L.u.w.s_wown; auto &s_pw = L.s_pw; auto &s_dmu = L.u.w.s_dmu; auto &s_rec = L.s_rec; auto &s_pcol = L.s_pcol;
    auto &s_np = L.s_np; auto &s_dc = L.u.w.s_dc; auto &s_pred = L.s_pred; auto &s_cpred = L.s_cpred; auto &s_stage = L.u.s_stage;
    auto &s_upair = L.s_upair; auto &s_ownsub = L.s_ownsub; auto &s_fk = L.s_fk; auto &s_fi = L.s_fi;
    double (*s_kown)[MID_ROWS] = (double (*)[MID_ROWS])&L.u.w.s_col[0][0][0];      // phase F / G: -K of the own rows, [k][row] (the inverse's patches are dead by then)
    static_assert(sizeof(L.u.w.s_col) >= sizeof(double) * MP * MID_ROWS, "s_kown fits the inverse's patches");
    int *const s_pair_obs = s_rec.pair_obs, *const s_pair_id = s_rec.pair_id, *const s_pair_state = s_rec.pair_state,
        *const s_urank = s_rec.urank, *const s_uid = s_rec.uid, *const s_cnt = s_rec.cnt, *const s_newid = s_rec.newid;
    static_assert(NPAIR <= 32, "RekfCtl::Rec holds a whole scan");
    double (*s_pu)[NUSP] = (double (*)[NUSP])L.s_big;                       // [sub-block row s][sub-block row s']: P(R, R)
    double (*s_sinv)[LDS_S] = (double (*)[LDS_S])L.s_big;                   // S^-1, row-major

#ifdef REKF_DEBUG_TIMING
    long long tqm[16]; int nqm = 0;
    const bool recm = (int)blockIdx.x == 1 + (FRONT ? A.front_in_mid : 0) && threadIdx.x == 0;      // (the second mid workgroup)
    const long long t_entrym = clock64(), w_entrym = wall_clock64();
#define MMARK() do { __builtin_amdgcn_sched_barrier(0); if (recm && nqm < 16) tqm[nqm++] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MMARK()
#endif
    // 512 threads = two teams of four waves (one of each per SIMD).  Waves 0..3 carry the critical chain -- sub-block
    // gather, W rows of S, S, its inverse -- waves 4..7 everything that only this workgroup's 16 rows / columns need
    // (H rows, own gathers, (H P)^T to HBM), off that chain; both meet at the barriers.
    RekfCtl *ctl = ctl_first;
#ifdef REKF_DEBUG_TIMING
    if (threadIdx.x == 0) {                         // (launch-level timeline: first entry of any workgroup, over the handle's life)
        const long long wnow = wall_clock64();
        if (ctl->dbg[28] == 0) ctl->dbg[28] = wnow;
        atomicMin((unsigned long long *)&ctl->dbg[28], (unsigned long long)wnow);
    }
#endif
    // ---- roles of the grid: [front end: A.front_in_mid workgroups][mid: A.n_mid][(idle up to A.dd_first)][downdate: the rest]
    if constexpr (DDROLE) {
        if (hd_dd_in_mid && (int)blockIdx.x >= hd_dd_first) {
            // the previous scan's downdate: waves 0..3 (the body is cut for 256 threads; a barrier counts the waves that have not ended)
            if (threadIdx.x < 256) {
                // (two device time stamps, 100 MHz, for the bench's roofline: when the role's first workgroup starts, when its last one ends)
                if (threadIdx.x == 0 && (int)blockIdx.x == A.dd_first) ctl->dbg[26] = wall_clock64();
                dd_body<64, true>(dp, k_mid_arena, (int)blockIdx.x - A.dd_first, A.dd_in_mid, &ctl->dd_queue[A.dd_par & 1], (int)blockIdx.x == A.dd_first);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (threadIdx.x == 0) atomicMax((unsigned long long *)&ctl->dbg[27], (unsigned long long)wall_clock64());
            }
            return;
        }
    }
    const int bxf = (int)blockIdx.x;
    if (FRONT && bxf < hd_front_in_mid) {
        // A scan's front end as the FIRST workgroups of this grid instead of a launch of its own (7.5 us + a kernel boundary in front
        // of k_mid, on the path every read-back caller waits for); the mid workgroups wait for its count below.  One-way: the front
        // role waits for nobody, its workgroups are dispatched first and the grid's first 256 workgroups are resident together.
        front_role<512>(d, A, bxf, A.front_in_mid, A.corr != 0);      // (A.corr: the pose block from RekfCtl::post_C9 -- the stored P is a scan behind)
        if constexpr (DDROLE) {
            // ... and then helps the downdate role out of the same queue
            if (A.dd_in_mid && threadIdx.x < 256) {
                __builtin_amdgcn_s_barrier();         // (waves 0..3: everybody is through with the front role's LDS)
                dd_body<64, true>(dp, k_mid_arena, -1, A.dd_in_mid, &ctl->dd_queue[A.dd_par & 1], false);
            }
        }
        return;
    }
    const int bx = bxf - (FRONT ? hd_front_in_mid : 0);          // this workgroup's number among the mid workgroups
    if constexpr (DDROLE && !FRONT) {
        if (hd_spec_front > 0 && bx >= hd_n_mid && bx < hd_n_mid + hd_spec_front) {
            // the NEXT scan's front end, speculatively (RekfCtl::spec): against the mean this launch's mid role starts from; nobody in
            // this launch waits for it.  Then it helps the downdate role.
            front_role<512, true>(d, An, bx - A.n_mid, A.spec_front, false);
            if (A.dd_in_mid && threadIdx.x < 256) {
                __builtin_amdgcn_s_barrier();
                dd_body<64, true>(dp, k_mid_arena, -1, A.dd_in_mid, &ctl->dd_queue[A.dd_par & 1], false);
            }
            return;
        }
    }
    if (bx >= hd_n_mid) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool steam = wave < 4;                  // S team; the other is the "own" team
    const int tt = tid & 255;                     // thread index within the team
    if (bx == 0 && tid == 511) { ctl->dd_queue[(hd_dd_par ^ 1) & 1] = 0u; ctl->dmmax[(hd_pred_ix + 1) & 3] = 0ull; }     // the next launch's tile queue (and mean-shift bound) start empty
    if (FRONT) {
        // every observation's result is in memory once the front end's count has reached the scan's target (each front workgroup
        // writes its result through, drains, then counts): wave 0 polls on one lane, takes the K results past this CU's L1 and
        // compacts them into the record for itself -- no compacting workgroup, no second hand-over
        if (tid < 64) {
            if (tid == 0) {
                unsigned spins = 0;
                while ((int)(__hip_atomic_load(&ctl->front_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - A.front_target) < 0 && ++spins < (1u << 22))
                    __builtin_amdgcn_s_sleep(2);
                if (spins >= (1u << 22)) atomicOr(&ctl->err, REKF_FLAG_STARVED);       // (never on an exclusive handle: rekf_api.hip)
            }
            __builtin_amdgcn_wave_barrier();
            const int kind = (tid < A.K) ? __hip_atomic_load(&ctl->obs_kind[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
            const int oidx = (tid < A.K) ? __hip_atomic_load(&ctl->obs_idx[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
            int n_rec = (d.n_known >= 0) ? d.n_known : ctl->n;                   // (the front role's own n: a scan behind a read-back knows it)
            if (AUGR && A.aug_in_mid && d.n_known < 0) {                           // (... or, while the filter grows, the previous scan's record: front_role)
                const RekfCtl::AugRec *arr = &ctl->augrec[(A.pred_slot ^ 1) & 1];
                n_rec = arr->n_before + 2 * arr->n2;
            }
            compact_record(&s_rec, ctl, kind, oidx, tid, A.K, n_rec, d.n_max, A.has_gps);
        }
        __syncthreads();
        MMARK();                                    // (in-grid front end through)
    }
    // the scan's match record first, UNCONDITIONALLY (a block step of a wide scan does not use it): a vector load that waits for no
    // scalar one, so the control block costs one memory round trip, not two
    constexpr int NREC = (int)(sizeof(RekfCtl::Rec) / sizeof(int));
    static_assert(NREC <= 512, "one load per thread");
    // MATCH GRID (RekfCtl::grid_state): no front end has run for this (host-predicted, whole) scan -- every workgroup matches it itself,
    // exactly: an observation can only match a landmark inside 0.6 m (cc:446), and every such landmark is in one of the 3 x 3 grid cells
    // around the observation's global point; the literal distances (cc:431-437) of those few candidates, first minimum in index order.
    // Wave w takes the observations w, w + 8, ...: all their bucket loads, then all their mean loads in flight together.
    const bool gm = GM && A.grid_match != 0 && A.pair0 < 0;
    // (what the phases behind the match need of the control block goes in flight FIRST: these loads then fly under the grid match instead of
    // costing a round trip of their own behind its barrier)
    int cp_uid_l = -1, cp_nu_l = -1;
    unsigned cp_scan_l = 0u;
    if (DDROLE && hd_corr && lane < 32) cp_uid_l = ctl->cp_uid[hd_corr_post][lane];
    if (DDROLE && hd_corr) { cp_nu_l = ctl->cp_nu[hd_corr_post]; cp_scan_l = ctl->cp_scan[hd_corr_post]; }
    double cpred_early = 0.0;                        // (s_cpred's source: the pending scan's (a, b) and its pose block after the update)
    if (DDROLE && hd_corr != 0 && tid >= 128 && tid < 128 + 11) {
        const int e = tid - 128;
        cpred_early = (e < 2) ? (hd_cpred != 0 ? ctl->pred[hd_corr_pred_ix].ab[e] : 0.0) : ctl->post_C9[hd_corr_post][e - 2];
    }
    if (GM && gm) {
#pragma clang fp contract(off)
#ifdef REKF_DEBUG_GRID
        MMARK();                                    // (g0: the grid match begins)
#endif
        const int nG = d.n_known, LG = (nG - 3) / 2;
        const double *muG = d.mu;
        const int grid_state_now = ctl->grid_state;                       // (in flight beside the bucket loads: looked at behind them)
        {
            constexpr int GQ = 4;                                         // K <= 32 observations over 8 waves
            // lane = (cell of the 3 x 3, slot of its bucket): count, landmark id and the landmark's current float32 mean (cc:431) in ONE
            // round trip -- all of this wave's observations' loads in flight together
            const int *gcnt = rekf_grid_cnt(d), *gid = rekf_grid_id(d);
            const float *gxy = rekf_grid_xy(d, d.grid_par);
            int cnt[GQ], cand[GQ];
            float ggx[GQ], ggy[GQ], flx[GQ], fly[GQ];
            const int nbk = lane / REKF_GRID_SLOTS, sl = lane - REKF_GRID_SLOTS * nbk;
            const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // (uniform on purpose: the observations then come by SCALAR loads out of the launch packet)
#pragma unroll
            for (int q = 0; q < GQ; ++q) {
                const int i = wave_u + 8 * q;
                cnt[q] = 0; cand[q] = -1; ggx[q] = 0.f; ggy[q] = 0.f; flx[q] = 0.f; fly[q] = 0.f;
                if (i < A.K) {
                    float gx0, gy0;
                    // (a grid-matched scan travels in the launch packet: A.obs directly -- one scalar round trip, not obs_ext's and then the data's)
                    obs_to_global(A.pre_pose[0], A.pre_pose[1], A.pre_pose[3], A.pre_pose[4], A.obs[2 * i], A.obs[2 * i + 1], gx0, gy0);
                    ggx[q] = gx0; ggy[q] = gy0;
                    if (lane < 9 * REKF_GRID_SLOTS) {
                        const int cx = (int)floorf(gx0) + (nbk % 3) - 1, cy = (int)floorf(gy0) + (nbk / 3) - 1;
                        const int hb = rekf_grid_hash(cx, cy, d.grid_mask), e = REKF_GRID_SLOTS * hb + sl;
                        cnt[q] = gcnt[hb]; cand[q] = gid[e];
                        const float2 xy = *(const float2 *)(gxy + 2 * (size_t)e);
                        flx[q] = xy.x; fly[q] = xy.y;
                    }
                }
            }
#ifdef REKF_DEBUG_GRID
            MMARK();                                // (g1: bucket loads issued)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            MMARK();                                // (g2: ... arrived)
#endif
#pragma unroll
            for (int q = 0; q < GQ; ++q) {
                const int i = wave_u + 8 * q;
                if (i < A.K) {                                             // (wave-uniform)
                    int kdm = 2, ixm = -1;
                    if (d.M_map > 0) map_match_wave(d, lane, ggx[q], ggy[q], kdm, ixm);               // cc:401-425 come first
                    const bool live = lane < 9 * REKF_GRID_SLOTS && sl < cnt[q] && cand[q] >= 0 && cand[q] < LG;
                    const float ex = ggx[q] - flx[q], ey = ggy[q] - fly[q];                           // cc:433
                    const double dx = (double)ex, dy = (double)ey;
                    const double d2 = dx * dx + dy * dy;
                    // cc:437-446 compare sqrt(d2) with 0.6 and take the first minimum in index order.  sqrt is monotone: away from the gate
                    // d2 < 0.36 decides the same, and ONE candidate inside needs no distance at all -- the usual case (a ballot, no
                    // reduction, no sqrt).  Only a candidate within rounding of the gate, or several inside, take the literal path.
                    const bool near_gate = live && fabs(d2 - 0.36) < 1e-12;
                    unsigned long long inside = __ballot(live && d2 < 0.36);
                    int best_j2 = -1;
                    if (__ballot(near_gate) != 0ull || __popcll(inside) > 1) {
                        const double dist = live ? sqrt(d2) : 1e300;                                  // cc:437, literally
                        inside = __ballot(live && dist < 0.6);
                        double best = 1e300;
                        while (inside) {
                            const int l = __builtin_ctzll(inside);
                            inside &= inside - 1ull;
                            const long long db = __double_as_longlong(dist);
                            const double dl = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(db >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)db, l));
                            const int jl = __builtin_amdgcn_readlane(cand[q], l);
                            if (best_j2 < 0 || dl < best || (dl == best && jl < best_j2)) { best = dl; best_j2 = jl; }
                        }
                    } else if (inside) best_j2 = __builtin_amdgcn_readlane(cand[q], __builtin_ctzll(inside));
                    int kd = kdm, ix = ixm;
                    if (kd == 2 && best_j2 >= 0) { kd = 1; ix = best_j2; }                            // cc:446
                    if (lane == 0) { s_fk[i] = kd; s_fi[i] = ix; }
                }
            }
        }
#ifdef REKF_DEBUG_GRID
        MMARK();                                    // (g3: this wave's observations matched)
#endif
        const bool grid_live = grid_state_now != 0;
        if (bx == 0 && tid == 0) {                     // (grid-matched scans / of those, by the full sweep: fire-and-forget counters)
            atomicAdd((unsigned long long *)&ctl->dbg[18], 1ull);
            if (!grid_live) atomicAdd((unsigned long long *)&ctl->dbg[19], 1ull);
        }
        if (!grid_live) {
            // the grid is not usable (a landmark has drifted too far from where it was binned, or a bucket ran over: the host rebuilds it or
            // gives it up): the exact match by the full sweep, one observation at a time by the whole workgroup -- slow, rare, the same result
            __syncthreads();
            for (int i = 0; i < A.K && i < 32; ++i) {
                float gx, gy;
                obs_to_global(A.pre_pose[0], A.pre_pose[1], A.pre_pose[3], A.pre_pose[4], rekf_obs(A, 2 * i), rekf_obs(A, 2 * i + 1), gx, gy);
                int kd, bj;
                rematch_obs(d, muG, LG, gx, gy, kd, bj);
                if (tid == 0) { s_fk[i] = kd; s_fi[i] = bj; }
            }
        }
        __syncthreads();
    }
    const bool cim = gm || (!FRONT && A.compact_in_mid != 0 && A.pair0 < 0);      // the front end's raw results instead (RekfFrontArgs::compact_in_mid)
    const int rec_raw = (!FRONT && !cim && tid < NREC) ? ((const int *)&ctl->rec[hd_pred_slot])[tid] : 0;
    const int cim_kind = (cim && tid < A.K && tid < 32) ? (gm ? s_fk[tid] : ctl->obs_kind[tid]) : -1,
              cim_idx = (cim && tid < A.K && tid < 32) ? (gm ? s_fi[tid] : ctl->obs_idx[tid]) : -1;
    // ... and the pending scan's WRITE-AHEAD CORRECTION (RekfCtl::cp_*, RekfDev::cp; phase G below): which landmarks it covers
#ifdef REKF_DEBUG_MID_FIRST
    MMARK();                                        // (x0: first loads issued)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MMARK();                                        // (x1: ... arrived)
#endif
    // the previous scan's augmentation, when it was deferred into this launch (RekfCtl::augrec, by scan parity): the state this scan
    // works on has n_before + 2 n2 rows, of which the last 2 n2 are being appended by workgroup 0 right now
    int ar_n = 0, ar_n2 = 0;
    if (AUGR && A.aug_in_mid) {
        const RekfCtl::AugRec *ar = &ctl->augrec[(A.pred_slot ^ 1) & 1];
        ar_n = __hip_atomic_load(&ar->n_before, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ar_n2 = __hip_atomic_load(&ar->n2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int n = (d.n_known >= 0) ? d.n_known : ((AUGR && A.aug_in_mid) ? ar_n + 2 * ar_n2 : ctl->n);
    const size_t ld = (size_t)d.ld;
    const int i0 = bx * MID_ROWS;
    // the host sizes the grid by its BOUND of n (it may run several scans ahead of the device, each of which can append K
    // reflectors): a workgroup past the real n has nothing to do -- and with one workgroup per CU (121 KB of LDS) a grid of more
    // than 256 would otherwise cost a second round of the whole inverse
    if (i0 >= n) return;
    if (AUGR && A.aug_in_mid && ar_n2 > 0) {
        // (rare: the previous scan met new reflectors.)  Workgroup 0 appends their covariance rows -- what k_augment would have done in
        // a launch of its own between the two scans -- and says so; everybody else waits for that before touching P: one poll loop on
        // one lane, one agent-scope acquire, a barrier (the grid's workgroups are resident together up to 256 x 16 rows, workgroup 0 is
        // dispatched first; the wait is bounded all the same and turns into the sticky STARVED error path: garbage, not a hang)
        if (bx == 0) {
            double *scr = L.s_big;
            const float *ao = ctl->augrec[(A.pred_slot ^ 1) & 1].obs;
            augment_rows(d, ar_n, ar_n2, A.obs_cov, (double (*)[6])scr, scr + 6 * REKF_MAX_OBS_DEV, scr + 6 * REKF_MAX_OBS_DEV + 9, 512,
                         [&](int k, float &rx, float &ry) { rx = ao[2 * k]; ry = ao[2 * k + 1]; });
            if (tid == 0) {
                ctl->n = ar_n + 2 * ar_n2;                                      // cc:360-363
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&ctl->aug_done, A.scan_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (tid == 0) {
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(&ctl->aug_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - A.scan_id) < 0 && ++spins < (1u << 22))
                __builtin_amdgcn_s_sleep(4);
            if (spins >= (1u << 22)) atomicOr(&ctl->err, REKF_FLAG_STARVED);
        }
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
    }
    const double *__restrict__ P = d.P;
    // (a host-predicted scan carries the predicted pose in the launch packet: no read of the control block for it)
    const bool hp = A.host_pred != 0;
    // (the front role inside this grid has written its Predict THROUGH to memory before it counted: coherent loads past this CU's L1)
    auto ctl_f64 = [&](const double *p) __attribute__((always_inline)) -> double {
        if constexpr (FRONT) return __longlong_as_double(__hip_atomic_load((const long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        else return *p;
    };
    const bool spec = DDROLE && !FRONT && hd_spec != 0;            // the record is speculative (RekfCtl::spec): no front end has run for this scan
    // (a speculative scan's Predict -- pose, (a, b), pose block -- was evaluated by workgroup 0 of the PREVIOUS scan's k_mid, at its end)
    const double *pp_src = spec ? ctl->pose_next[hd_pred_slot] : ctl->pose_pred;
    const double pose[5] = {hp ? A.pre_pose[0] : ctl_f64(&pp_src[0]), hp ? A.pre_pose[1] : ctl_f64(&pp_src[1]), hp ? A.pre_pose[2] : ctl_f64(&pp_src[2]),
                            hp ? A.pre_pose[3] : ctl_f64(&pp_src[3]), hp ? A.pre_pose[4] : ctl_f64(&pp_src[4])};
    const bool pending = (FRONT || spec || gm) ? true : ctl->pose_pending != 0;      // (the in-grid front role sets it beside us: a host-predicted scan always has one)
    // (a grid-matched scan had no front end to leave its Predict in the control block: the scan's downdate and the next scan's correction
    // read it there -- workgroup 0 writes what the launch packet carries)
    if (gm && bx == 0 && tid >= 192 && tid < 192 + 11) {
        const int e = tid - 192;
        ((double *)&ctl->pred[A.pred_ix & 3])[e] = (e < 2) ? A.pre_ab[e] : A.pre_C9[e - 2];
    }
    // (the match grid's share of this workgroup's rows: where each landmark was binned, and its entry -- whose mean phase F keeps current)
    const bool grid_row = d.grid_bucket && tid < MID_ROWS && bx * MID_ROWS + tid >= 3 && bx * MID_ROWS + tid < d.n_max;
    const float grid_p0_own = grid_row ? d.grid_p0[bx * MID_ROWS + tid - 3] : 0.f;
    const int grid_slot_own = grid_row ? rekf_grid_slot(d)[(bx * MID_ROWS + tid - 3) >> 1] : -1;
    const bool first = bx == 0;
    // a speculative scan: its per-observation results with their margins (lane = observation)
    int sp_kind = -1, sp_idx = -1;
    double sp_d1 = 0.0, sp_d2 = 0.0, sp_dm1 = 1e300, sp_rng = 0.0, sp_pose0 = 0.0, sp_pose1 = 0.0, sp_pose2 = 0.0, sp_dmm = 0.0;
    bool sp_rec_ok = false;
    if (spec) {
        // (everything the proof needs goes in flight now: behind the barrier only arithmetic is left)
        const RekfCtl::Spec *sq = &ctl->spec[hd_pred_slot];
        if (lane < hd_K) {
            sp_kind = sq->kind[lane & 31]; sp_idx = sq->idx[lane & 31]; sp_d1 = sq->d1[lane & 31]; sp_d2 = sq->d2[lane & 31];
            if (d.M_map > 0) sp_dm1 = sq->dm1[lane & 31];
            const double px = (double)rekf_obs(A, 2 * lane), py = (double)rekf_obs(A, 2 * lane + 1);
            sp_rng = sqrt(px * px + py * py);
        }
        sp_pose0 = sq->pose[0]; sp_pose1 = sq->pose[1]; sp_pose2 = sq->pose[2];
        sp_dmm = __longlong_as_double((long long)ctl->dmmax[(hd_pred_ix + 3) & 3]);
        sp_rec_ok = sq->scan == hd_scan_id && sq->n == ((d.n_known >= 0) ? d.n_known : -1);
    }
    // the scan's pending Predict (RekfCtl::pred): applied to the gathered P in phase D.  (a, b) = 0 and the pose block as gathered
    // when nothing is pending (later block steps of a wide scan: the first step's downdate has committed it)
    // (through LDS, not registers: eleven uniform doubles held from here to phase D cost this 512-thread kernel its residency)
    const bool do_pred = A.apply_pred != 0;
    if (do_pred && tid >= 64 && tid < 64 + 11) {                          // ab[0], ab[1], C9[0..8]
        // (with the front role in this grid the control block's copy is being written beside us: a host-predicted scan carries the values)
        const int e = tid - 64;
        s_pred[e] = ((FRONT || gm) && hp) ? (e < 2 ? A.pre_ab[e] : A.pre_C9[e - 2]) : ctl_f64(&((const double *)&ctl->pred[hd_pred_ix])[e]);
    }
    // ... and what is PENDING on the stored P (A.corr): the previous scan's (a, b) and its pose block after the update
    const bool corr = DDROLE && hd_corr != 0, cpred = corr && hd_cpred != 0;     // (MODE 1, the two-launch chain of a filter that can still grow: never)
    if (corr && tid >= 128 && tid < 128 + 11) s_cpred[tid - 128] = cpred_early;

    // ---- A: the scan's matched pairs.  Whole scan (pair0 < 0): the record the front end left.  Block step of a wide scan (pair0 >= 0): the
    // pairs [pair0, pair0 + stride) of the record k_compact_wide wrote, state pairs first, then map pairs, compacted here.
    if (tid < 64 && A.pair0 >= 0) {
        const int M = ctl->n_state, MMtot = M + ctl->n_map;
        const int left = MMtot - A.pair0;
        const int cnt = left < 0 ? 0 : (left < A.pair_stride ? left : A.pair_stride);
        const int gpi = A.pair0 + lane;
        const bool live = lane < cnt, st = live && gpi < M;
        int ob = 0, id = 0;
        if (live) {
            const int *pp = st ? &ctl->state_pairs[2 * gpi] : &ctl->map_pairs[2 * (gpi - M)];
            ob = pp[0]; id = pp[1];
        }
        int rk = 0;
        const int key = st ? id : 0x7fffffff;
#pragma unroll
        for (int q = 0; q < 2 * NPAIR; ++q) {
            const int oq = __builtin_amdgcn_readlane(key, q);
            rk += (oq < key || (oq == key && q < lane)) ? 1 : 0;
        }
        int NSl = M - A.pair0;
        NSl = NSl < 0 ? 0 : (NSl < cnt ? NSl : cnt);
        int sorted, nu_l; bool dup;
        const int ur = distinct_ranks(key, rk, NSl, lane, sorted, dup, nu_l);
        const unsigned long long D = __ballot(dup);
        if (lane < NSl && lane < NPAIR && !dup) s_uid[lane - __popcll(D & ((1ull << lane) - 1ull))] = sorted;
        if (live && lane < NPAIR) {
            s_pair_obs[lane] = ob; s_pair_id[lane] = id; s_pair_state[lane] = st ? 1 : 0;
            if (st) s_urank[lane] = ur;
        }
        if (lane == 0) {
            const bool gps = A.has_gps && cnt > 0 && A.pair0 + cnt == MMtot;     // the pose rows ride on the block step that holds the last pairs
            const int m = (cnt > 0) ? 2 * cnt + (gps ? 3 : 0) : 0;
            s_cnt[0] = cnt; s_cnt[1] = m; s_cnt[2] = (m + 15) & ~15; s_cnt[3] = NSl; s_cnt[4] = gps ? 1 : 0;
            s_cnt[5] = (A.pair0 + A.pair_stride >= A.K) ? ctl->n_new : 0;       // the LAST block step appends the new reflectors' means
            s_rec.nu = nu_l;
        }
    } else if (A.pair0 < 0) {
        // whole scan: the record the front end left (its last workgroup compacted the results, front_role) -- one load, one LDS store
        if (cim) { if (tid < 64) compact_record(&s_rec, ctl, cim_kind, cim_idx, tid, A.K, n, d.n_max, A.has_gps); }
        else if (!FRONT && tid < NREC) ((int *)&s_rec)[tid] = rec_raw;            // (FRONT: compacted above)
    }
#ifdef REKF_DEBUG_MID_FIRST
    MMARK();                                        // (x4: wave 0 through the compaction)
#endif
    __syncthreads();
    MMARK();                                        // 0: compaction done
    if (spec) {
        // ---- the speculative record, proved or repaired.  The front role matched this scan against the pose (sq->pose) and the means
        // scan t - 1's launch STARTED from; since then that scan's update has moved the pose (by what pose - sq->pose says: both are
        // known) and every reflector mean by at most dmmax.  An observation's distance to ANY reflector has therefore changed by at most
        //     delta = |dxy| + range * |dtheta| + sqrt(2) dmmax + (float32 roundings of the two evaluations),
        // and its decision stands if it has that much room: matched (d1 < 0.6): d1 + delta < 0.6 and d2 - d1 > 2 delta (the nearest stays
        // the nearest, first-index ties cannot arise); unmatched: d1 - delta > 0.6.  Every wave decides for itself from the same numbers.
        bool okv = true;
        {
#pragma clang fp contract(off)
            const double dxp = pose[0] - sp_pose0, dyp = pose[1] - sp_pose1;
            double dth = pose[2] - sp_pose2;
            dth = dth - 6.283185307179586 * rint(dth / 6.283185307179586);
            const double dmm = sp_dmm;
            if (lane < A.K) {
                const double rng = sp_rng;
                // dg: how far the observation's global point can have moved (pose shift + the float32 roundings of the two evaluations);
                // the state branch adds the reflectors' own shift; the map branch (fixed points) scales dg by the map's Lipschitz bound
                const double dg = sqrt(dxp * dxp + dyp * dyp) + rng * fabs(dth) + 3e-7 * (fabs(pose[0]) + fabs(pose[1]) + rng + 1.0);
                const double delta = dg + 1.5 * dmm;
                const double dmap = d.map_lip * dg + 1e-12;
                // with a pre-loaded map an observation reaches the state branch only if NO map point is inside 0.05 (cc:401-425 come first)
                const bool map_free = d.M_map == 0 || (d.map_lip >= 0.0 && sp_dm1 - dmap > 0.05);
                if (sp_kind == 1) okv = map_free && (sp_d1 + delta < 0.6) && (sp_d2 - sp_d1 > 2.0 * delta);
                else if (sp_kind == 2) okv = map_free && (sp_d1 - delta > 0.6);
                else if (sp_kind == 0) okv = d.map_lip >= 0.0 && (sp_d1 + dmap < 0.05) && (sp_d2 - sp_d1 > 2.0 * dmap);
                else okv = false;
                if (!(delta == delta) || !(dmap == dmap)) okv = false;
            }
        }
        const bool rec_ok = sp_rec_ok && d.n_known == n;
        unsigned long long bad = __ballot(!okv);
        if (!rec_ok) bad = (A.K >= 64) ? ~0ull : ((1ull << A.K) - 1ull);
        if (first && tid == 0) { ctl->dbg[20] += 1; if (bad) ctl->dbg[21] += 1; }      // (speculative scans / those with observations re-matched)
        if (!bad && first && tid < 64) {
            // the record stands as the front role left it -- without the capacity flag (compact_record, raise_flag): raised here, by the scan itself
            const int n_kind2 = __popcll(__ballot(lane < A.K && sp_kind == 2));
            if (lane == 0 && n_kind2 > (d.n_max - n) / 2) atomicOr(&ctl->err, REKF_FLAG_CAPACITY);
        }
        if (bad) {
            // (rare) the exact match of the observations that did not pass, one at a time by the whole workgroup; then the record again
            if (tid < 32) { s_fk[tid] = sp_kind; s_fi[tid] = sp_idx; }
            for (int i = 0; i < A.K && i < 32; ++i) {
                if ((bad >> i) & 1ull) {
                    float gx, gy;
                    obs_to_global(pose[0], pose[1], pose[3], pose[4], rekf_obs(A, 2 * i), rekf_obs(A, 2 * i + 1), gx, gy);
                    int kd, bj;
                    rematch_obs(d, d.mu, (n - 3) / 2, gx, gy, kd, bj);
                    if (tid == 0) { s_fk[i] = kd; s_fi[i] = bj; }
                }
            }
            __syncthreads();
            if (tid < 64) compact_record(&s_rec, ctl, (tid < A.K && tid < 32) ? s_fk[tid] : -1, (tid < A.K && tid < 32) ? s_fi[tid] : -1, tid, A.K, n, d.n_max, A.has_gps);
            __syncthreads();
        }
    }
#ifdef REKF_DEBUG_MID_FINE
    MMARK();                                        // (f0: speculative record proved)
#endif
    const int MM = s_cnt[0], m = s_cnt[1], m_pad = s_cnt[2], NS = s_cnt[3];
    const bool gps_rows = s_cnt[4] != 0;
    // the pending scan's write-ahead correction serves this scan when it covers exactly this scan's landmarks
    bool use_cp = false;
    {
        const int nu0 = (NS > 0) ? s_rec.nu : 0;
        const bool eq = lane >= 32 || lane >= nu0 || cp_uid_l == s_uid[lane];
        use_cp = DDROLE && hd_corr != 0 && A.pair0 < 0 && cp_scan_l == hd_corr_scan && cp_nu_l == nu0 && __ballot(eq) == ~0ull;
    }
    if (first && A.pair0 < 0) {
        // ReflectorMatchResult for the getters (and n_new / m for the kernels behind this one), out of the record
        const int Mm = s_cnt[6], N2r = s_cnt[5];
        if (tid < NS) { ctl->state_pairs[2 * tid] = s_pair_obs[tid]; ctl->state_pairs[2 * tid + 1] = s_pair_id[tid]; }
        else if (tid < MM) { ctl->map_pairs[2 * (tid - NS)] = s_pair_obs[tid]; ctl->map_pairs[2 * (tid - NS) + 1] = s_pair_id[tid]; }
        if (tid >= 64 && tid < 64 + N2r) ctl->new_ids[tid - 64] = s_newid[tid - 64];
        if (tid == 128) {
            ctl->K = s_cnt[7]; ctl->n_state = NS; ctl->n_map = Mm; ctl->n_new = N2r;
            ctl->m = m; ctl->m_pad = m_pad;
        }
        // EARLY n (rekf_api.hip, struct rekf): the n this scan leaves, for the host that plans the next scan's launch -- now, not at the kernel's end
        if (d.early && tid == 130) host_slot_store(d.early, (double)(n + 2 * N2r), d.early_seq, 0);
        if (DDROLE && A.cp_write) {                     // which landmarks this scan's write-ahead correction covers (phase G)
            const int nu0 = (NS > 0) ? s_rec.nu : 0;
            if (tid >= 256 && tid < 256 + 32) ctl->cp_uid[d.post_slot & 1][tid - 256] = (tid - 256 < nu0) ? s_uid[tid - 256] : -1;
            if (tid == 288) { ctl->cp_nu[d.post_slot & 1] = nu0; ctl->cp_scan[d.post_slot & 1] = A.scan_id; }
        }
        // ... and what this scan's augmentation needs, should it be deferred into the next scan's k_mid (RekfCtl::augrec)
        if ((AUGW || d.aug_write) && A.K <= REKF_MAX_OBS_DEV) {
            RekfCtl::AugRec *aw = &ctl->augrec[A.pred_slot & 1];
            if (tid >= 192 && tid < 192 + N2r) {
                const int lid = s_newid[tid - 192];
                aw->obs[2 * (tid - 192)] = rekf_obs(A, 2 * lid); aw->obs[2 * (tid - 192) + 1] = rekf_obs(A, 2 * lid + 1);
            }
            if (tid == 129) { aw->n_before = n; aw->n2 = N2r; }
        }
    }
    // The means of the scan's NEW reflectors (cc:323-342: the observation through the UPDATED pose, float32-rounded) are written
    // here, by workgroup 0 behind its pose commit, not by k_augment: the next scan's match may then run before k_augment has
    // appended their covariance rows (lazy downdate, rekf_api.hip).  s_np = the committed pose; call with the whole workgroup.
    auto append_new_means = [&]() __attribute__((always_inline)) {
        __syncthreads();
        const int N2w = s_cnt[5];
        if (tid < N2w) {
#pragma clang fp contract(off)
            const double x = s_np[0], y = s_np[1], th = s_np[2];
            const double sn = sin(th), cs = cos(th);                    // cc:323-324
            const int local_id = (A.pair0 >= 0) ? ctl->new_ids[tid] : s_newid[tid];   // cc:338 (block steps: k_compact_wide's record)
            float gx, gy;
            obs_to_global(x, y, cs, sn, rekf_obs(A, 2 * local_id), rekf_obs(A, 2 * local_id + 1), gx, gy);
            d.mu_out[n + 2 * tid] = (double)gx;                         // cc:341-342
            d.mu_out[n + 2 * tid + 1] = (double)gy;
            // ... and the match grid takes them in where they stand (RekfCtl::grid_state; nobody reads the grid in this launch any more:
            // every workgroup matched at its start)
            if (d.grid_bucket && ctl->grid_state) {
                grid_insert(d, ctl, (n - 3) / 2 + tid, gx, gy);
                if (ctl->grid_overflow && d.grid_note) host_slot_store(d.grid_note, 2.0, (int)A.scan_id, 0);
            }
        }
    };
    // The NEXT scan's Predict, when that scan is in the speculation pipeline (its launch packet An came with this launch): Predict's scalar
    // part (cc:154-206) is evaluated HERE, by workgroup 0 at its tail -- the front end's own source (front_role, device-predicted path) on
    // the pose this scan commits (s_np) and the pose block it leaves (s_cpred[2..10]) -- while the other workgroups write their share of
    // the write-ahead correction: the next k_mid starts with one load of it.  Called on EVERY path that ends a scan, the one without a
    // matched observation included (round-5 advice: that path returned in front of it, and the next scan read a Predict two scans old).
    auto next_scan_predict = [&]() __attribute__((always_inline)) {
        if constexpr (DDROLE && !FRONT) {
            if (A.spec_front > 0) {
                const int ns = (A.pred_slot ^ 1) & 1;
                if (tid == 0) {
#pragma clang fp contract(off)
                    const double th = s_np[2] + An.vt[2] * An.dt;
                    double sn, cs;
                    sincos(th, &sn, &cs);
                    ctl->pose_next[ns][3] = cs; ctl->pose_next[ns][4] = sn;
                    ctl->pose_next[ns][2] = atan2(sn, cs);                     // the wrapped heading (cc:181 / :205)
                }
                if (tid == 64) {
#pragma clang fp contract(off)
                    Motion mo;
                    double C9[9];
                    for (int q = 0; q < 9; ++q) C9[q] = s_cpred[2 + q];
                    motion_terms(An, s_np[2], mo);
                    ctl->pose_next[ns][0] = s_np[0] + mo.d[0]; ctl->pose_next[ns][1] = s_np[1] + mo.d[1];
                    corner_predict(C9, 3, mo);
                    RekfCtl::Pred *pr = &ctl->pred[(A.pred_ix + 1) & 3];
                    pr->ab[0] = mo.a; pr->ab[1] = mo.b;
                    for (int q = 0; q < 9; ++q) pr->C9[q] = C9[q];
                }
            }
        }
    };
    double *const post_out = ctl->post_C9[d.post_slot & 1];
    // The mean is double-buffered: other workgroups read landmark means from d.mu (phase B) while this one is already
    // done, so the updated rows go to d.mu_out and the host swaps the two pointers behind this launch.
    if (m == 0) {                                   // nothing matched: commit the predicted pose (cc:234 + Predict), no update
        if (tid < MID_ROWS && i0 + tid < n) {
            const int i = i0 + tid;
            const double pp = (i == 0) ? pose[0] : ((i == 1) ? pose[1] : pose[2]);      // no dynamic indexing of pose[]
            const double vv = (pending && i < 3) ? pp : d.mu[i];
            d.mu_out[i] = vv;
            if (i >= 3 && grid_slot_own >= 0) rekf_grid_xy(d, d.grid_par ^ 1)[2 * grid_slot_own + ((i - 3) & 1)] = (float)vv;
            if (first && i < 3) s_np[i] = vv;
        }
        if (pending && first && tid == 0) ctl->pose_pending = 0;
        if (first) {
            // nothing to update: the pose block is the predicted one (or, in a later block step of a wide scan, what the previous
            // step's downdate left in memory -- or will leave: the pending scan's block)
            if (tid < 9) {
                const int pi = tid % 3, pj = tid / 3, hi = pi > pj ? pi : pj, lo = pi > pj ? pj : pi;
                const double v9 = do_pred ? s_pred[2 + hi + 3 * lo] : (corr ? ctl->post_C9[A.corr_post & 1][hi + 3 * lo] : rekf_plower(P, (int)ld, hi, lo));
                post_out[tid] = v9;
                s_cpred[2 + tid] = v9;          // (for the next scan's Predict below; the pending scan's block, if it was there, is dead)
                if (d.pub) host_slot_store(d.pub + 3 + tid, v9, d.pub_seq, 0);
            }
            append_new_means();                 // (a barrier inside: s_np and the pose block are complete behind it)
            next_scan_predict();
            if (d.pub) {                        // (the early publisher, see the end of the kernel)
                if (tid < 3) host_slot_store(d.pub + tid, s_np[tid], d.pub_seq, 0);
                if (tid == 3) host_slot_store(d.pub + 12, (double)(n + 2 * s_cnt[5]), d.pub_seq,
                                              __hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
        }
        // the downdate runs without looking at the control block when the host knows n: give it zeros to add
        for (int e = tid; e < 16 * REKF_PANEL_COLS; e += 512) {
            const int cidx = e & 15, r = e >> 4, c = i0 + cidx;
            d.HPt[c + (size_t)r * ld] = 0.0;
            d.Kn[c + (size_t)r * ld] = 0.0;
        }
        return;
    }
    // (m_pad <= MP: the host picked NBR from its bound 2K(+3) of m)
#ifdef REKF_DEBUG_MID_FINE
    MMARK();                                        // (f1: in front of phase C)
#endif

    // ---- C (issued before B: it needs only the match record): P(R, R) and P(own, R).  R in ascending global order: sub-block row
    // s < 3 is global row s, s = 3 + 2 u + e is row e of the u-th distinct matched landmark.  P is stored as its LOWER triangle and ONE
    // SCAN BEHIND: element (hi, lo), hi >= lo, of the covariance this scan sees is
    //     stored(hi, lo)  [+ a stored(hi, 2) | + b stored(hi, 2)   for lo = 0 | 1, hi >= 3: the pending Predict]
    //                     + sum_k HPt(lo, k) Kn(hi, k)              the pending downdate, dd_body's own MFMA chain over k
    // and the pose block is the pending scan's RekfCtl::post_C9 -- the very bits that scan's downdate stores (it runs beside us, from
    // this buffer into the other).
    // C0: the stored values, as rounds 3-4 gathered them -- the sub-block's lower block-triangle as 2 x 2 blocks (row slot u, column
    // slot v <= u; slots: rows {0,1}, row {2}, then the distinct landmarks), column slot major so that neighbouring lanes hit neighbouring
    // rows of one column, two 16-byte loads per block, mirrored on the way into LDS; own rows x R as row pairs -- and, for the correction,
    // the rows R of the pending Kn / HPt panels into LDS ([k][s], lane = slot, wave = k: 16-byte loads down a column of the panel).
    const int nu = (NS > 0) ? s_rec.nu : 0;
    const int nus = 3 + 2 * nu, nbu = (nus + 15) >> 4, nrs = nu + 2;
    auto ug_of = [&](int s) __attribute__((always_inline)) -> int {          // global row of sub-block row s (s < nus)
        return (s < 3) ? s : 3 + 2 * s_uid[(s - 3) >> 1] + ((s - 3) & 1);
    };
    auto slot_row = [&](int u) __attribute__((always_inline)) -> int { return (u < 2) ? 2 * u : 3 + 2 * s_uid[u - 2]; };      // first global row of a slot
    const unsigned ldb = (unsigned)d.ld * 8u;                              // bytes per column of P (byte offsets fit 32 bits: ld^2 * 8 < 4 GiB)
    const double *__restrict__ cHPt = dp.HPt;
    const double *__restrict__ cKn = dp.Kn;
    const int ckc = corr ? dp.kc_ub : 0, cnk = ckc >> 2;                     // columns / k-steps of the pending panels (their columns [m, 64) are zero)
    const int g4 = lane >> 4, c16 = lane & 15;
    constexpr int NRS = NPAIR + 2, NBLK = NRS * (NRS + 1) / 2, PS_IT = (NBLK + 255) / 256;
    constexpr int NBLK_MAX = NBU * (NBU + 1) / 2, SB_MAX = (NBLK_MAX + 7) / 8;   // 16 x 16 blocks of the lower block-triangle; per wave (all eight take some)
    constexpr int OI_MAX = (NBU + 2 + 3) / 4;                                 // (block, orientation) items per own-team wave
    constexpr int PW_IT = (LT::NUS * 8 + 255) / 256;
    constexpr int ST_IT = 8;                                                  // stage: k = wave + 8 j
    // own team: its (block, orientation) items.  Orientation 1: own row = `hi` (the Kn side, MFMA column c), sub-block row = `lo`;
    // 2: own row = `lo` (the HPt side, MFMA row g + 4 r), sub-block row = `hi`; a block takes the orientations its global rows call for
    int it_bv[OI_MAX], it_or[OI_MAX];
    double own_op[OI_MAX][16];                                                // the own rows' operand of each item, straight from the panel
    double so[OI_MAX][16];                                                    // ... and the sub-block rows' operand, out of the staged panels
    {
#pragma unroll
        for (int q = 0; q < OI_MAX; ++q) { it_bv[q] = -1; it_or[q] = 0; }
    }
    // the panels' rows R -> LDS (all eight waves): lane = slot, k = wave + 8 j
    v2du stv[2][ST_IT];
    const bool slow_corr = corr && !use_cp;          // the correction is computed here (C1) instead of taken from the write-ahead panel
    if (first && tid == 0 && corr) { ctl->dbg[22] += 1; if (slow_corr) ctl->dbg[23] += 1; }      // (scans that met a pending downdate / computed its correction themselves)
    const bool st_live = slow_corr && lane < nrs;
    const int st_row = st_live ? slot_row(lane) : 0;
    if (slow_corr) {
#pragma unroll
        for (int j = 0; j < ST_IT; ++j) {
            const int k = wave + 8 * j;
            stv[0][j] = (v2du){0, 0}; stv[1][j] = (v2du){0, 0};
            if (st_live && k < ckc) {
                stv[0][j] = *(const v2du *)(cHPt + (size_t)st_row + (size_t)k * ld);
                stv[1][j] = *(const v2du *)(cKn + (size_t)st_row + (size_t)k * ld);
            }
        }
    }
#ifdef REKF_DEBUG_MID_FINE
    MMARK();                                        // (f2: stage loads issued (slow path), tables set up)
#endif
    v2du ps[PS_IT][2];
    const double *__restrict__ cpP = d.cp + (size_t)(A.corr_post & 1) * REKF_CP_LD * REKF_CP_LD;
    constexpr int FP_IT = (LT::NUS * (REKF_CP_LD / 2) + 255) / 256;           // the write-ahead panel, 16 bytes per thread and pass
    v2d fpv[FP_IT];
    int blk_u[PS_IT], blk_v[PS_IT];                                           // row slot u, column slot v of this thread's blocks (-1: none)
    v2d pw[PW_IT];
    if (steam && use_cp) {
        // the sub-block AFTER the pending scan's update is in that scan's write-ahead panel (phase G of its k_mid): 36 KB, read linearly
#pragma unroll
        for (int it = 0; it < FP_IT; ++it) {
            const int e = tt + 256 * it, row = e / (REKF_CP_LD / 2), c2 = 2 * (e - row * (REKF_CP_LD / 2));
            fpv[it] = (v2d){0, 0};
            if (row < nus && c2 <= row) fpv[it] = *(const v2d *)(cpP + (size_t)row * REKF_CP_LD + c2);
        }
    } else if (steam) {
        const int nblk = nrs * (nrs + 1) / 2;
#pragma unroll
        for (int it = 0; it < PS_IT; ++it) {
            const int t = tt + 256 * it;
            const int tc = (t < nblk) ? t : nblk - 1;                       // clamped: a thread past the end refetches the last block
            // column slot v of block tc in the column-major enumeration of the lower block-triangle: v columns hold v nrs - v (v - 1) / 2 blocks
            const float bq = 2.0f * (float)nrs + 1.0f;
            int v = (int)((bq - sqrtf(fmaxf(bq * bq - 8.0f * (float)tc, 0.0f))) * 0.5f);
            v = max(0, min(nrs - 1, v));
            auto c0 = [&](int vv) __attribute__((always_inline)) { return vv * nrs - (vv * (vv - 1)) / 2; };   // blocks in columns < vv
            while (v > 0 && c0(v) > tc) --v;
            while (v + 1 < nrs && c0(v + 1) <= tc) ++v;
            const int u = v + (tc - c0(v));
            blk_u[it] = (t < nblk) ? u : -1; blk_v[it] = v;
            const char *rp = (const char *)(P + slot_row(u)) + (unsigned)slot_row(v) * ldb;
            ps[it][0] = *(const v2du *)rp;                                  // rows (r, r+1) of column c
            ps[it][1] = *(const v2du *)(rp + ldb);                          // ... of column c + 1
        }
    } else {
        // sub-block row s -> its global row, for s = lane and s = 64 + lane, once: the loop fetches it with a shuffle
        int colA = 0, colB = 0;
        if (lane < nus) colA = ug_of(lane);
        if (64 + lane < nus) colB = ug_of(64 + lane);
        const int pr = tid & 7, sub = (tid >> 3) & 7;                       // 8 sub-block rows x 8 own row pairs per wave instruction
        const int kc0 = __builtin_amdgcn_readfirstlane(wave & 3);
        const int i = i0 + 2 * pr;
#pragma unroll
        for (int it = 0; it < PW_IT; ++it) {
            int kc = 8 * (kc0 + 4 * it) + sub;
            kc = (kc < nus) ? kc : nus - 1;
            const int cA = __shfl(colA, kc & 63, 64), cB = __shfl(colB, kc & 63, 64);
            const int cc = (kc < 64) ? cA : cB;
            // (i, cc) and (i + 1, cc), each from below the diagonal
            const double *e0 = (i >= cc) ? P + (size_t)i + (size_t)cc * ld : P + (size_t)cc + (size_t)i * ld;
            const double *e1 = (i + 1 >= cc) ? P + (size_t)(i + 1) + (size_t)cc * ld : P + (size_t)cc + (size_t)(i + 1) * ld;
            pw[it].x = *e0; pw[it].y = *e1;
        }
        if (corr) {
            const int ow = wave & 3;
            int cnt_it = 0;
            for (int bv = 0; bv < nbu; ++bv) {
                const int s_lo = 16 * bv, s_hi = (16 * bv + 15 < nus) ? 16 * bv + 15 : nus - 1;
                const int umin = ug_of(s_lo), umax = ug_of(s_hi);
#pragma unroll
                for (int o = 1; o <= 2; ++o) {
                    const bool want = (o == 1) ? (umin <= i0 + MID_ROWS - 1) : (umax > i0);
                    if (want) {
                        if ((cnt_it & 3) == ow) {
#pragma unroll
                            for (int q = 0; q < OI_MAX; ++q) if (q == (cnt_it >> 2)) { it_bv[q] = bv; it_or[q] = o; }
                        }
                        ++cnt_it;
                    }
                }
            }
        }
    }
    MMARK();                                        // 1: gathers issued

    // ---- B: H rows (cc:248-304, gps.cc:305-332), thread = row (own team)
    if (!steam && tt < MP) {
        const int r = tt, p = r >> 1, rr = r & 1;
        double hr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int col = -1;
        if (p < MM) {
            const HPair h = make_hpair(d, A, pose, s_pair_obs[p], s_pair_id[p], s_pair_state[p]);
            col = h.col;
            if (rr == 0) { hr[0] = h.a0[0]; hr[1] = h.a0[1]; hr[2] = h.a0[2]; hr[3] = h.b0[0]; hr[4] = h.b0[1]; hr[5] = h.q0; hr[6] = h.dz0; }
            else { hr[0] = h.a1[0]; hr[1] = h.a1[1]; hr[2] = h.a1[2]; hr[3] = h.b1[0]; hr[4] = h.b1[1]; hr[5] = h.q1; hr[6] = h.dz1; }
            if (col < 0) { hr[3] = 0; hr[4] = 0; }                          // map rows carry no landmark block (cc:285-303)
        } else if (gps_rows && r >= 2 * MM && r < 2 * MM + 3) {            // pose rows: unit vectors, fixed noise, wrapped yaw innovation
#pragma clang fp contract(off)
            const int k = r - 2 * MM;
            hr[0] = (k == 0); hr[1] = (k == 1); hr[2] = (k == 2);
            hr[5] = (k == 2) ? 0.017 * 0.017 : 0.05 * 0.05;
            const double e0 = A.gps[0] - pose[0], e1 = A.gps[1] - pose[1], e2 = yaw_innovation(A.gps[2] - pose[2]);
            hr[6] = (k == 0) ? e0 : ((k == 1) ? e1 : e2);
        }
        if (A.pair0 > 0 && r < m) {
            // A later block step of a wide scan.  The joint update of y = H x + v (v uncorrelated between rows: Q is
            // diagonal, cc:276,302, gps.cc:312-316) equals block-sequential updates with the SAME linearisation when each
            // step's innovation is taken against the mean the earlier steps left: dz_b - H_b (mu_now - mu_lin).
#pragma clang fp contract(off)
            double dth = d.mu[2] - pose[2];
            dth = atan2(sin(dth), cos(dth));
            double corr_dz = hr[0] * (d.mu[0] - pose[0]);
            corr_dz += hr[1] * (d.mu[1] - pose[1]);
            corr_dz += hr[2] * dth;
            if (col >= 0) {
                corr_dz += hr[3] * (d.mu[col] - d.mu_lin[col]);
                corr_dz += hr[4] * (d.mu[col + 1] - d.mu_lin[col + 1]);
            }
            hr[6] = hr[6] - corr_dz;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) s_coef[8 * r + q] = hr[q];
        if (rr == 0) {
            s_pcol[p] = (col >= 0) ? 3 + 2 * s_urank[p] : -1;               // (the pair's landmark as sub-block rows)
            if (col >= 0) s_upair[s_urank[p]] = p;                           // (two observations on one landmark: either)
        }
    }
    // stored values -> LDS.  s_pu is [sub-block row][sub-block row], both triangles: a block goes in twice, as fetched and transposed.
    // Slot 1 stands for row / column 2 ALONE: its second row (3) is fetched but not used.
    auto slot_s = [](int u) __attribute__((always_inline)) -> int { return (u < 2) ? 2 * u : 2 * u - 1; };     // first sub-block row of a slot
    if (steam && use_cp) {
        // elements (row, c2), (row, c2 + 1) of the lower triangle, mirrored; the pose block by value (the pending scan's post_C9)
#pragma unroll
        for (int it = 0; it < FP_IT; ++it) {
            const int e = tt + 256 * it, row = e / (REKF_CP_LD / 2), c2 = 2 * (e - row * (REKF_CP_LD / 2));
            if (row < nus && c2 <= row) {
                double v0 = fpv[it].x, v1 = fpv[it].y;
                if (row < 3) { v0 = s_cpred[2 + row + 3 * c2]; v1 = (c2 + 1 <= row) ? s_cpred[2 + row + 3 * (c2 + 1)] : 0.0; }
                s_pu[row][c2] = v0; s_pu[c2][row] = v0;
                if (c2 + 1 <= row) { s_pu[row][c2 + 1] = v1; s_pu[c2 + 1][row] = v1; }
            }
        }
    } else if (steam) {
#pragma unroll
        for (int it = 0; it < PS_IT; ++it) {
            const int u = blk_u[it], v = blk_v[it];
            if (u >= 0) {
                const int su = slot_s(u), sv = slot_s(v);
                const bool ua = u != 1, ve = v != 1;
                double b00 = ps[it][0].x, b10 = ps[it][0].y, b01 = ps[it][1].x, b11 = ps[it][1].y;   // b[a][e] = P(r + a, c + e)
                if (u == v) b01 = b10;                                      // a diagonal block: (r, r+1) lies above the diagonal
                s_pu[su][sv] = b00;
                if (ua) s_pu[su + 1][sv] = b10;
                if (ve) s_pu[su][sv + 1] = b01;
                if (ua && ve) s_pu[su + 1][sv + 1] = b11;
                if (u != v) {
                    s_pu[sv][su] = b00;
                    if (ua) s_pu[sv][su + 1] = b10;
                    if (ve) s_pu[sv + 1][su] = b01;
                    if (ua && ve) s_pu[sv + 1][su + 1] = b11;
                }
            }
        }
    }
    if (slow_corr) {
        if (st_live) {
            const int s0 = slot_s(lane);
#pragma unroll
            for (int j = 0; j < ST_IT; ++j) {
                const int k = wave + 8 * j;
                if (k < ckc) {
                    s_stage[0][k][s0] = stv[0][j].x; s_stage[1][k][s0] = stv[1][j].x;
                    if (lane != 1) { s_stage[0][k][s0 + 1] = stv[0][j].y; s_stage[1][k][s0 + 1] = stv[1][j].y; }
                }
            }
        }
        __syncthreads();
        MMARK();                                    // (stored values and the pending panels' rows in LDS)
        // C1: the pending correction of the SUB-BLOCK, 16 x 16 blocks, on all eight waves: MFMA rows <-> sub-block rows 16 bv + g + 4 r
        // (the `lo` side: the HPt operand), columns <-> 16 bu + c (the `hi` side: Kn); ONE chain over k from zero, in k order -- dd_body's
        // accumulation -- two blocks' chains interleaved.  Every element has one owner: it reads the stored value, adds, writes both
        // triangles.  (This workgroup's own rows take theirs later, off the critical chain: under the inverse, phase E.)
        {
            const int nblk16 = nbu * (nbu + 1) / 2, chunk = (nblk16 + 7) >> 3;
            int sb_hi[SB_MAX], sb_lo0[SB_MAX];           // per block: this lane's `hi` sub-block row (-1: none), the block's first `lo` row (-1: no block)
            int sb_bu[SB_MAX];
            v4d acc[SB_MAX];
#pragma unroll
            for (int b = 0; b < SB_MAX; ++b) {
                const int t = wave * chunk + b;
                const bool live = b < chunk && t < nblk16;
                int bu = nbu - 1, rem = live ? t : 0;    // block t of the enumeration (bu descending, bv ascending)
                while (rem > bu) { rem -= bu + 1; --bu; }
                const int sB = 16 * bu + c16;
                sb_bu[b] = bu;
                sb_hi[b] = (live && sB < nus) ? sB : -1;
                sb_lo0[b] = live ? 16 * rem : -1;
                acc[b] = (v4d){0, 0, 0, 0};
            }
#pragma unroll
            for (int b0 = 0; b0 < SB_MAX; b0 += 2) {
                // both blocks' operands of all 16 k-steps first (the LDS reads issue together), then the two chains, interleaved.  No branch
                // in here: a wave without a block in this position multiplies whatever block 0's addresses hold and never looks at the sums;
                // k-steps past the pending panels' columns take zero operands (as the downdate reads the panels' zero columns)
                double oa[2][16], ob[2][16];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int b = (b0 + h < SB_MAX) ? b0 + h : b0;
                    const int sa = (sb_lo0[b] >= 0 ? sb_lo0[b] : 0) + c16, sbb = 16 * sb_bu[b] + c16;
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const double ra = s_stage[0][4 * q + g4][sa], rb = s_stage[1][4 * q + g4][sbb];
                        oa[h][q] = (q < cnk) ? ra : 0.0; ob[h][q] = (q < cnk) ? rb : 0.0;
                    }
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int b = b0 + h;
                        if (b < SB_MAX) {
#ifndef REKF_ABL_SC1
                            acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(oa[h][q], ob[h][q], acc[b], 0, 0, 0);
#else
                            acc[b][0] += oa[h][q] * ob[h][q];
#endif
                        }
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < SB_MAX; ++b) {
                if (sb_hi[b] >= 0) {
#pragma clang fp contract(off)
                    const int sB = sb_hi[b];
                    double vv[4];
                    const double raw2 = s_pu[sB][2];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int sj = sb_lo0[b] + g4 + 4 * r;
                        vv[r] = (sj <= sB) ? s_pu[sB][sj] : 0.0;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int sj = sb_lo0[b] + g4 + 4 * r;
                        if (sj <= sB) {
                            double v = vv[r];
                            if (sB < 3) v = s_cpred[2 + sB + 3 * sj];                 // the pose block: the pending scan's post_C9 (hi + 3 lo)
                            else {
                                if (cpred && sj < 2) v = v + s_cpred[sj] * raw2;      // the pending Predict: columns 0, 1 against column 2
                                v = v + acc[b][r];
                            }
                            s_pu[sB][sj] = v;
                            s_pu[sj][sB] = v;
                        }
                    }
                }
            }
        }
    }
    lds_barrier();                                  // (LDS only: the own team's gathers may still be in flight -- nobody needs them before phase E)
    MMARK();                                        // 2: H rows and the sub-block in LDS
    if (!steam) {
        // the own rows' stored values -> LDS (read by the own team itself, behind the next barrier: own_correct / own_w under the inverse)
        const int pr = tid & 7, sub = (tid >> 3) & 7;
        const int kc0 = __builtin_amdgcn_readfirstlane(wave & 3);
#pragma unroll
        for (int it = 0; it < PW_IT; ++it) {
            const int kc = 8 * (kc0 + 4 * it) + sub;
            if (kc < nus) *(v2d *)&s_pw[kc][2 * pr] = pw[it];
        }
    }

    // (phase E multiplies the landmark rows of W of a CLAMPED pair by the zero coefficients of rows that have no landmark block:
    // with no state pair at all -- a scan that matched the pre-loaded map only -- that pair does not exist, and whatever the
    // last kernel left in LDS there must not be a NaN)
    if (NS == 0 && tid < MP) { s_wcp[0][tid][0] = 0.0; s_wcp[0][tid][1] = 0.0; }
    // ---- D: form W (rows of S, own rows) and (H P)^T (own columns) out of LDS.
    // Same operation order as k_gather: v = p0 h0; v += p1 h1; v += p2 h2; v += pl0 g0; v += pl1 g1
    const int nq = m_pad / 2;                        // row pairs, pad rows included (their H rows are zero)
    if (steam) {
        // items (slot, q): slot < NS = state pair (its two landmark rows), slot NS = rows 0,1, slot NS+1 = row 2 (its second
        // value is row 3: computed, never stored).  q = tid mod NPAIR, slot = tid / NPAIR + (256 / NPAIR) pass
        const int q = tt & (NPAIR - 1);
        const bool qlive = q < nq;
        const int pc = qlive ? s_pcol[q] : -1;
        const bool has_col = pc >= 0;                 // a state pair: its landmark's columns are sub-block rows pc, pc + 1
        const int kq = has_col ? pc : 0;              // clamped: without a landmark block the two extra terms are multiplied by zeros
        double h0[5], h1[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) { h0[k] = s_coef[16 * q + k]; h1[k] = s_coef[16 * q + 8 + k]; }
        if (!has_col) { h0[3] = 0; h0[4] = 0; h1[3] = 0; h1[4] = 0; }
        constexpr int NRS = NPAIR + 2, SL_STEP = 256 / NPAIR, SL_PASS = (NRS + SL_STEP - 1) / SL_STEP;
        const int nrs = NS + 2;
        int rs2[SL_PASS];
#pragma unroll
        for (int pass = 0; pass < SL_PASS; ++pass) {
            const int slot = tt / NPAIR + SL_STEP * pass, sl = (slot < nrs) ? slot : nrs - 1;
            rs2[pass] = (sl < NS) ? 3 + 2 * s_urank[sl] : ((sl == NS) ? 0 : 2);          // first sub-block row of the slot
        }
#pragma unroll
        for (int pass = 0; pass < SL_PASS; ++pass) {
            const int slot = tt / NPAIR + SL_STEP * pass;
            v2d a01 = *(const v2d *)&s_pu[rs2[pass]][0], b01 = *(const v2d *)&s_pu[rs2[pass] + 1][0];
            double a2 = s_pu[rs2[pass]][2], b2 = s_pu[rs2[pass] + 1][2];
            v2d al = *(const v2du *)&s_pu[rs2[pass]][kq], bl = *(const v2du *)&s_pu[rs2[pass] + 1][kq];
            if (do_pred) {
                // the scan's Predict, G P G^T + V restricted to the sub-block (G = I + a e0 e2^T + b e1 e2^T): a landmark row takes
                // P(r, 0) + a P(r, 2) and P(r, 1) + b P(r, 2) (the same single operations the old covariance pass did in memory);
                // rows 0, 1 take the predicted pose block and, against a landmark column c, P(0, c) + a P(2, c) / P(1, c) + b P(2, c);
                // row 2 takes the predicted pose block and keeps its landmark columns
#pragma clang fp contract(off)
                const double pa = s_pred[0], pb = s_pred[1];
                const double *pC9 = s_pred + 2;
                if (slot < NS) {
                    a01.x = a01.x + pa * a2; a01.y = a01.y + pb * a2;
                    b01.x = b01.x + pa * b2; b01.y = b01.y + pb * b2;
                } else if (slot == NS) {
                    const v2d r2 = *(const v2du *)&s_pu[2][kq];                         // P(2, c), P(2, c + 1)
                    a01.x = pC9[0]; a01.y = pC9[3]; a2 = pC9[6];                        // row 0 of the predicted block: (0,0) (0,1) (0,2)
                    b01.x = pC9[1]; b01.y = pC9[4]; b2 = pC9[7];                        // row 1
                    al.x = al.x + pa * r2.x; al.y = al.y + pa * r2.y;
                    bl.x = bl.x + pb * r2.x; bl.y = bl.y + pb * r2.y;
                } else {
                    a01.x = pC9[2]; a01.y = pC9[5]; a2 = pC9[8];                        // row 2
                }
            }
            double v0x = a01.x * h0[0], v0y = b01.x * h0[0], v1x = a01.x * h1[0], v1y = b01.x * h1[0];
            v0x += a01.y * h0[1]; v0y += b01.y * h0[1]; v1x += a01.y * h1[1]; v1y += b01.y * h1[1];
            v0x += a2 * h0[2]; v0y += b2 * h0[2]; v1x += a2 * h1[2]; v1y += b2 * h1[2];
            if (has_col) {                           // (kept as a select: adding +0.0 products would turn a -0.0 sum into +0.0)
                v0x += al.x * h0[3]; v0y += bl.x * h0[3]; v1x += al.x * h1[3]; v1y += bl.x * h1[3];
                v0x += al.y * h0[4]; v0y += bl.y * h0[4]; v1x += al.y * h1[4]; v1y += bl.y * h1[4];
            }
            if (slot < nrs && qlive) {
                if (slot < NS) { *(v2d *)&s_wcp[slot][2 * q][0] = (v2d){v0x, v0y}; *(v2d *)&s_wcp[slot][2 * q + 1][0] = (v2d){v1x, v1y}; }
                else if (slot == NS) { *(v2d *)&s_wc0[0][2 * q] = (v2d){v0x, v1x}; *(v2d *)&s_wc0[1][2 * q] = (v2d){v0y, v1y}; }
                else *(v2d *)&s_wc0[2][2 * q] = (v2d){v0x, v1x};
            }
        }
    }
    // ---- the own team's share, OFF the critical chain: it runs under the inverse (phase E: gj_invert_blocks calls the idle waves back between
    // its barriers).  First the pending correction of P(own rows, R) -- the MFMA chains, then every element's owner reads the stored value,
    // adds, writes -- then W for the own rows and (H P)^T to HBM.
    if (!steam && corr) {                              // (the operands of this workgroup's own rows' correction: in flight across phase D)
#pragma unroll
        for (int q = 0; q < OI_MAX; ++q) {
            // the own rows' side: 16 consecutive rows of one panel; the sub-block rows' side: rows R of the other
            const int sc = 16 * (it_bv[q] >= 0 ? it_bv[q] : 0) + c16;
            const int gsr = (it_bv[q] >= 0 && sc < nus) ? ug_of(sc) : 0;
            const double *po = ((it_or[q] == 2) ? cHPt : cKn) + (size_t)(i0 + c16) + (size_t)g4 * ld;
            const double *pr2 = ((it_or[q] == 2) ? cKn : cHPt) + (size_t)gsr + (size_t)g4 * ld;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                own_op[q][k] = 0.0; so[q][k] = 0.0;
                if (k < cnk && it_bv[q] >= 0) { own_op[q][k] = po[(size_t)(4 * k) * ld]; so[q][k] = pr2[(size_t)(4 * k) * ld]; }
            }
        }
    }
    auto own_correct = [&]() __attribute__((always_inline)) {
        if (!corr) return;
        v4d acc[OI_MAX];
#pragma unroll
        for (int q = 0; q < OI_MAX; ++q) acc[q] = (v4d){0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
#pragma unroll
            for (int q = 0; q < OI_MAX; ++q) {
                // (orientation by select of the operands, not by branch; an empty position's sums are never looked at)
                const double oa1 = (it_or[q] == 2) ? own_op[q][k] : so[q][k], ob1 = (it_or[q] == 2) ? so[q][k] : own_op[q][k];
#ifndef REKF_ABL_OWNC1
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(oa1, ob1, acc[q], 0, 0, 0);
#else
                acc[q][0] += oa1 * ob1;
#endif
            }
        }
#pragma unroll
        for (int q = 0; q < OI_MAX; ++q) {
            if (it_bv[q] >= 0) {
#pragma clang fp contract(off)
                int hi_g[4], lo_g[4], s_dst[4], row_dst[4];
                bool ok[4];
                double vv[4], r2[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (it_or[q] == 1) {
                        const int sj = 16 * it_bv[q] + g4 + 4 * r;
                        const int gj = (sj < nus) ? ug_of(sj) : 0x7fffffff;
                        hi_g[r] = i0 + c16; lo_g[r] = gj; s_dst[r] = sj; row_dst[r] = c16; ok[r] = sj < nus && gj <= hi_g[r];
                    } else {
                        const int sx = 16 * it_bv[q] + c16, io = i0 + g4 + 4 * r;
                        const int gs = (sx < nus) ? ug_of(sx) : -1;
                        hi_g[r] = gs; lo_g[r] = io; s_dst[r] = sx; row_dst[r] = g4 + 4 * r; ok[r] = gs > io;
                    }
                    vv[r] = 0.0; r2[r] = 0.0;
                    if (ok[r]) {
                        vv[r] = s_pw[s_dst[r]][row_dst[r]];
                        // (hi, 2) of this workgroup's gather: orientation 1: own row hi against sub-block row 2; 2 (workgroup 0, own rows 0, 1): sub-block row against own row 2
                        r2[r] = (it_or[q] == 1) ? s_pw[2][row_dst[r]] : s_pw[s_dst[r]][2];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (ok[r]) {
                        double v = vv[r];
                        if (hi_g[r] < 3) v = s_cpred[2 + hi_g[r] + 3 * lo_g[r]];
                        else {
                            if (cpred && lo_g[r] < 2) v = v + s_cpred[lo_g[r]] * r2[r];
                            v = v + acc[q][r];
                        }
                        s_pw[s_dst[r]][row_dst[r]] = v;
                    }
                }
            }
        }
    };
    auto own_w = [&]() __attribute__((always_inline)) {
        // own rows: item (row pair pr, q), q = tt / 8 (+ 32 per pass)
        const int pr = tt & 7;
#pragma unroll
        for (int pass = 0; pass < (NPAIR + 31) / 32; ++pass) {
            const int q2 = (tt >> 3) + 32 * pass;
            if (q2 < nq) {
                const int pc2 = s_pcol[q2];
                const bool hc = pc2 >= 0;
                const int c = i0 + 2 * pr;
                v2d p0 = *(const v2d *)&s_pw[0][2 * pr], p1 = *(const v2d *)&s_pw[1][2 * pr], p2 = *(const v2d *)&s_pw[2][2 * pr];
                v2d l0 = {0, 0}, l1 = {0, 0};
                if (hc) { l0 = *(const v2d *)&s_pw[pc2][2 * pr]; l1 = *(const v2d *)&s_pw[pc2 + 1][2 * pr]; }
                if (do_pred) {                        // the scan's Predict on this workgroup's rows (see the S team above)
#pragma clang fp contract(off)
                    const double pa = s_pred[0], pb = s_pred[1];
                    const double *pC9 = s_pred + 2;
                    if (c >= 3) {
                        p0.x = p0.x + pa * p2.x; p1.x = p1.x + pb * p2.x;
                        p0.y = p0.y + pa * p2.y; p1.y = p1.y + pb * p2.y;
                    } else if (c == 0) {              // rows 0, 1 (workgroup 0): the predicted block; landmark columns against row 2's
                        p0.x = pC9[0]; p0.y = pC9[1]; p1.x = pC9[3]; p1.y = pC9[4]; p2.x = pC9[6]; p2.y = pC9[7];
                        if (hc) {
                            const double r20 = s_pw[pc2][2], r21 = s_pw[pc2 + 1][2];
                            l0.x = l0.x + pa * r20; l0.y = l0.y + pb * r20;
                            l1.x = l1.x + pa * r21; l1.y = l1.y + pb * r21;
                        }
                    } else {                          // c == 2: row 2 (the predicted block), row 3 (an ordinary row)
                        p0.y = p0.y + pa * p2.y; p1.y = p1.y + pb * p2.y;
                        p0.x = pC9[2]; p1.x = pC9[5]; p2.x = pC9[8];
                    }
                }
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int r = 2 * q2 + rr;
                    const double *h = s_coef + 8 * r;
                    double vx = p0.x * h[0], vy = p0.y * h[0];
                    vx += p1.x * h[1]; vy += p1.y * h[1];
                    vx += p2.x * h[2]; vy += p2.y * h[2];
                    if (hc) {
                        vx += l0.x * h[3]; vy += l0.y * h[3];
                        vx += l1.x * h[4]; vy += l1.y * h[4];
                    }
                    if (c >= n) vx = 0.0;
                    if (c + 1 >= n) vy = 0.0;
                    *(v2d *)&s_wown[r][2 * pr] = (v2d){vx, vy};
                    // (H P)^T(c, r) = W(c, r) (symmetric P): 8 lanes store 128 contiguous bytes of column r
                    store_wt2(&d.HPt[(size_t)c + (size_t)r * ld], (rekf_v2d){vx, vy});
                }
            }
        }
        // columns [m_pad, 64) of HPt / Kn are kept zero for the downdate (whichever KC it runs with)
        for (int e = tt; e < 16 * (REKF_PANEL_COLS - m_pad); e += 256) {
            const int cidx2 = e & 15, r = m_pad + (e >> 4), c2 = i0 + cidx2;
            d.HPt[c2 + (size_t)r * ld] = 0.0;
            d.Kn[c2 + (size_t)r * ld] = 0.0;
        }
        // which of this workgroup's rows are sub-block rows (phase G); one wave: its LDS operations execute in order
        if (DDROLE && wave == 4) {
            if (lane < MID_ROWS) s_ownsub[lane] = -1;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int sx = lane + 64 * h;
                if (sx >= 3 && sx < nus) { const int gr = ug_of(sx) - i0; if (gr >= 0 && gr < MID_ROWS) s_ownsub[gr] = sx; }      // (the pose rows' elements are the pose block: taken by value)
            }
        }
    };
    lds_barrier();                                  // (LDS only: the own team's operand loads stay in flight)
    MMARK();                                        // 3: W rows of S in LDS

    // ---- E: S = H W + Q in the C layout (wave w = block column w), inverse, S^-1 -> LDS
    const int nbr = m_pad >> 4;
    const int g = lane >> 4, c = lane & 15;
    {
        v4d S[NBR];
#pragma unroll
        for (int bi = 0; bi < NBR; ++bi) S[bi] = (v4d){0, 0, 0, 0};
        const int w = wave, j = 16 * w + c;
        if (w < nbr) {
            // straight-line on purpose (all LDS reads of a block row issue together): rows without a landmark block have
            // zero coefficients there (phase B), so their two extra terms are computed against a clamped pair and add 0
            const int last_pair = (NS > 0) ? NS - 1 : 0;
            const double w0 = s_wc0[0][j], w1 = s_wc0[1][j], w2 = s_wc0[2][j];
#pragma unroll
            for (int bi = 0; bi < NBR; ++bi) {
                if (bi < nbr) {
                    v2d ha01[4], ha2b0[4], b1q[4], wl[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * bi + g + 4 * r, pi = (i >> 1) < NS ? (i >> 1) : last_pair;
                        ha01[r] = *(const v2d *)(s_coef + 8 * i); ha2b0[r] = *(const v2d *)(s_coef + 8 * i + 2);
                        b1q[r] = *(const v2d *)(s_coef + 8 * i + 4);
                        wl[r] = *(const v2d *)&s_wcp[pi][j][0];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * bi + g + 4 * r;
                        double v = ha01[r].x * w0;
                        v += ha01[r].y * w1;
                        v += ha2b0[r].x * w2;
                        v += ha2b0[r].y * wl[r].x;
                        v += b1q[r].x * wl[r].y;
                        if (i == j) v += b1q[r].y;
                        if (i >= m || j >= m) v = (i == j) ? 1.0 : 0.0;
                        S[bi][r] = v;
                    }
                }
            }
        }
        MMARK();                                    // 4: S built
        const bool bad = gj_invert_blocks<NBR>(S, nbr, w, g, c, s_col, s_leaf[w & 3], [&](int slot) __attribute__((always_inline)) {
            if (slot == 0) own_correct();             // (every element of s_pw has one owner; the barrier behind slot 0 orders it in front of own_w)
            else if (slot == 1) own_w();
        }
#ifdef REKF_DEBUG_GJ
        , [&]() __attribute__((always_inline)) { MMARK(); }
#endif
        );
        MMARK();                                    // 5: inverted
        if (w < nbr) {
            if (bad && lane == 0 && first) atomicOr(&ctl->err, REKF_FLAG_SINGULAR);
#pragma unroll
            for (int bi = 0; bi < NBR; ++bi) {
                if (bi < nbr) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) s_sinv[16 * bi + g + 4 * r][j] = S[bi][r];
                }
            }
        }
    }
    __syncthreads();
    MMARK();                                        // 6: S^-1 in LDS

    // ---- F: K = W S^-1 for this workgroup's 16 rows, transposed like k_gain (MFMA rows <-> j, columns <-> i)
    {
        const int idx = lane & 15, kq = lane >> 4;
        for (int jt = wave; jt < NBR; jt += 4) {
            const int j0 = 16 * jt;
            double part = 0.0, pc0 = 0.0, pc1 = 0.0, pc2 = 0.0;
            if (jt < nbr) {
                // two accumulation chains (a dependent MFMA costs ~100 cycles, an independent one 64), operands of a whole
                // 16-row block of k read before its MFMAs issue
                v4d acc = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
                for (int kb = 0; kb < NBR; ++kb) {
                    if (kb < nbr) {
                        double av[4], bv[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) { const int k = 16 * kb + 4 * q + kq; av[q] = s_sinv[k][j0 + idx]; bv[q] = s_wown[k][idx]; }
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], acc, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], acc1, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], acc, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], acc1, 0, 0, 0);
                    }
                }
                acc = acc + acc1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = j0 + kq + 4 * r;                               // D row = column of K
                    store_wt(&d.Kn[(i0 + idx) + (size_t)j * ld], -acc[r]);
                    if (DDROLE) s_kown[j][idx] = -acc[r];                        // (phase G)
                    part += acc[r] * s_coef[8 * j + 6];                          // K(i, j) (z - zhat)(j)
                    if (first) {                                                 // (K H P)(i, jc) = sum_j K(i, j) W(jc, j), jc = 0..2
                        pc0 += acc[r] * s_wown[j][0]; pc1 += acc[r] * s_wown[j][1]; pc2 += acc[r] * s_wown[j][2];
                    }
                }
                part += __shfl_xor(part, 16, 64);
                part += __shfl_xor(part, 32, 64);
                if (first) {
                    pc0 += __shfl_xor(pc0, 16, 64); pc0 += __shfl_xor(pc0, 32, 64);
                    pc1 += __shfl_xor(pc1, 16, 64); pc1 += __shfl_xor(pc1, 32, 64);
                    pc2 += __shfl_xor(pc2, 16, 64); pc2 += __shfl_xor(pc2, 32, 64);
                }
            }
            if (kq == 0) s_dmu[jt & 3][idx] = part;
            if (first && kq == 0) { s_dc[jt & 3][0][idx] = pc0; s_dc[jt & 3][1][idx] = pc1; s_dc[jt & 3][2][idx] = pc2; }
        }
    }
    __syncthreads();
    MMARK();                                        // 7: K stored
    if (tid < MID_ROWS) {
        const int i = i0 + tid;
        double adm = 0.0;                               // |mu_new - mu_old| of a landmark row (the next scan's speculation bound, RekfCtl::dmmax)
        if (i < n) {
            double dm = 0.0;
#pragma unroll
            for (int jt = 0; jt < NBR; ++jt) dm += s_dmu[jt][tid];
            const double pp = (i == 0) ? pose[0] : ((i == 1) ? pose[1] : pose[2]);
            const double base = (pending && i < 3) ? pp : d.mu[i];
            double v = base + dm;
            if (i == 2) v = atan2(sin(v), cos(v));                               // cc:307
            d.mu_out[i] = v;
            if (i >= 3) adm = fabs(dm);
            if (i >= 3 && grid_slot_own >= 0) rekf_grid_xy(d, d.grid_par ^ 1)[2 * grid_slot_own + ((i - 3) & 1)] = (float)v;     // (the entry's mean, for the NEXT launch: cc:431's float32)
            // the match grid stays exact only while every landmark is within REKF_GRID_DRIFT of where it was binned
            if (i >= 3 && d.grid_bucket && fabsf((float)v - grid_p0_own) > d.grid_drift) {
                if (ctl->grid_state) { ctl->grid_state = 0; if (d.grid_note) host_slot_store(d.grid_note, 1.0, (int)A.scan_id, 0); }
            }
            if (i == 0) ctl->pose_pending = 0;
            if (first && i < 3) s_np[i] = v;
        }
        if (DDROLE) {
            adm = vmax_f64(adm, __shfl_xor(adm, 1, 64)); adm = vmax_f64(adm, __shfl_xor(adm, 2, 64));
            adm = vmax_f64(adm, __shfl_xor(adm, 4, 64)); adm = vmax_f64(adm, __shfl_xor(adm, 8, 64));
            if (tid == 0 && adm > 0.0) atomicMax(&ctl->dmmax[A.pred_ix & 3], (unsigned long long)__double_as_longlong(adm));
        }
    }
    if (first) {
        // the pose block after the update, P'(i, j) - (K H P)(i, j), lower element for both halves (RekfCtl::post_C9): to the control
        // block: the downdate's tile (0, 0) stores it
        if (tid >= 64 && tid < 64 + 9) {
#pragma clang fp contract(off)
            const int e = tid - 64, pi = e % 3, pj = e / 3, hi = pi > pj ? pi : pj, lo = pi > pj ? pj : pi;
            double khp = 0.0;
#pragma unroll
            for (int jt = 0; jt < NBR; ++jt) khp += s_dc[jt][lo][hi];
            const double base = do_pred ? s_pred[2 + hi + 3 * lo] : s_pw[lo][hi];
            const double v9 = base - khp;
            post_out[e] = v9;
            s_cpred[2 + e] = v9;                // (the pending scan's block is dead by now: this scan's, for the next scan's Predict below)
            if (d.pub) host_slot_store(d.pub + 3 + e, v9, d.pub_seq, 0);
        }
        append_new_means();                     // (a barrier inside: s_np is complete behind it)
        next_scan_predict();
        // The EARLY publisher (d.pub set on this launch: the caller has been reading the pose back after its scans, rekf_api.hip): pose,
        // block, n and flags go to the host from here, a kernel before the downdate -- at the price of 0.8 us at the end of this kernel
        // (it ends when the PCIe writes are through).  Otherwise the downdate's first workgroup publishes, at its start.
        if (d.pub) {
            if (tid < 3) host_slot_store(d.pub + tid, s_np[tid], d.pub_seq, 0);
            if (tid == 3) host_slot_store(d.pub + 12, (double)(n + 2 * s_cnt[5]), d.pub_seq,
                                          __hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
    }
    // ---- G: this scan's WRITE-AHEAD CORRECTION.  The next scan meets the stored P one scan behind and needs, of this scan's rank-m downdate,
    // what falls on ITS sub-block -- almost always this scan's (RekfCtl::cp_*).  Element (hi, lo) of it, sum_k HPt(lo, k) Kn(hi, k), is
    // computed here by the workgroup that owns row hi: Kn(hi, :) is its own (phase F), HPt(lo, :) = W(lo, :) sits in LDS for every lo in R
    // (phase D built it for S) -- the downdate's own MFMA chain over k, so the sum is the one the downdate will add.  One 16 x 16 block
    // product per wave and block of sub-block rows at or below this workgroup's.
    if (DDROLE && A.cp_write) {
        const int s_hi = s_ownsub[c16];
        // (blocks whose first row lies above every sub-block row this workgroup owns have nothing for it)
        int smax = s_hi;
        smax = max(smax, __shfl_xor(smax, 1, 64)); smax = max(smax, __shfl_xor(smax, 2, 64));
        smax = max(smax, __shfl_xor(smax, 4, 64)); smax = max(smax, __shfl_xor(smax, 8, 64));
        const int bvw = wave;
        if (smax >= 0 && bvw < nbu && 16 * bvw <= smax) {
            const int sa = 16 * bvw + c16;               // this lane's `lo` row of the HPt operand
            const int ua = (sa >= 3 && sa < nus) ? s_upair[(sa - 3) >> 1] : 0, ea = (sa - 3) & 1;
            v4d acc = {0, 0, 0, 0};
            double oa[16], ob[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k = 4 * q + g4;
                const int kk = (k < m_pad) ? k : 0;
                const double wa = (sa < 3) ? s_wc0[sa < 3 ? sa : 0][kk] : s_wcp[ua][kk][ea];
                const double kb = s_kown[kk][c16];
                oa[q] = (k < m_pad && sa < nus) ? wa : 0.0; ob[q] = (k < m_pad) ? kb : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(oa[q], ob[q], acc, 0, 0, 0);
            double *cpo = d.cp + (size_t)(d.post_slot & 1) * REKF_CP_LD * REKF_CP_LD;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma clang fp contract(off)
                // the element AFTER this scan's update, in the downdate's own order: (stored [+ this scan's Predict on columns 0, 1]) + sum_k
                const int sj = 16 * bvw + g4 + 4 * r;
                if (s_hi >= 0 && sj <= s_hi) {
                    double base = s_pw[sj][c16];
                    if (do_pred && sj < 2) base = base + s_pred[sj] * s_pw[2][c16];
                    store_wt(&cpo[(size_t)s_hi * REKF_CP_LD + sj], base + acc[r]);
                }
            }
        }
    }
#if defined(REKF_DEBUG_TIMING) && !defined(REKF_DEBUG_DD2)
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        atomicMax((unsigned long long *)&ctl->dbg[30], (unsigned long long)wall_clock64());      // last exit of a mid workgroup
        if (first) ctl->dbg[29] = wall_clock64();                                                 // workgroup 0's exit
        if (bx == A.n_mid - 1) ctl->dbg[25] = w_entrym;                                           // the last mid workgroup's entry
    }
    if (recm) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ctl->dbg[6] = clock64() - t_entrym;
        ctl->dbg[5] = wall_clock64() - w_entrym;
        ctl->dbg[4] = w_entrym; ctl->dbg[31] = wall_clock64();
        ctl->dbg[7] = nqm;
        for (int i = 0; i < nqm; ++i) ctl->dbg[8 + i] = tqm[i] - t_entrym;
    }
#endif
}

// ----------------------------------------------------------------------------
// The rank-m downdate  P(i,j) += sum_k Kn(i,k) HPt(j,k)   (P <- P - K (H P), cc:308): k_downdate2 below.
// 64 x 64 tiles of P; per tile the Kn row panel and the HPt row panel sit in LDS ([k][64] doubles each) and every
// wave owns a 32 x 32 sub-tile = 2 x 2 v_mfma_f64_16x16x4_f64 accumulators.  The MFMA is evaluated transposed (MFMA
// M <-> j, N <-> i) and MFMA tile t of a pair covers the interleaved rows i = base + 2 idx + t, so that each lane holds
// two adjacent rows of one column: 16-byte global accesses, 256 contiguous bytes per 16 lanes, and ONE ds_read_b128 per
// operand per k-step feeds both tiles, conflict-free on the linear [k][64] LDS image.
// ----------------------------------------------------------------------------
#define DT 64
#define DD_WG_PER_CU 1

// ----------------------------------------------------------------------------
// k_downdate2<KC>: the rank-m downdate when the scan's innovation fits one k-chunk, m_pad <= KC <= 64 (the host picks
// KC from its bound 2K(+3) of m; k_gather / k_gain keep the columns [m, KC) of HPt / Kn zero).  Round-2 pipeline.
//
// What changed against the register-staged version above (which needed 18.5 us at C3 although its
// P traffic alone takes 11 us and its MFMAs alone 9 us -- the two did not overlap: store burst, LDS
// panel write and barrier sat between the MFMA loops of consecutive tiles):
//   * the Kn / HPt panels go global -> LDS by DMA (global_load_lds_dwordx4, 1 KiB per wave
//     instruction = two k-rows of a panel, which is exactly the linear [k][64] LDS image): no staging
//     registers, no ds_write pass;
//   * a panel is fetched only when its tile row / column CHANGES: a workgroup's tiles run down one
//     tile column, so HPt(J) is loaded once and only Kn(I) streams (half the panel traffic);
//   * the accumulators start at ZERO and P is added at the end of the tile (P + sum_k, the order the
//     reference's `sigma - K*H*sigma` has), so a tile's P block is not needed until its MFMA loop is
//     over: it is requested during the PREVIOUS tile;
//   * P + acc lands in the registers that held P: the stores of tile t are issued from there in the
//     first half of tile t+1's MFMA loop, then the same registers receive the P block of tile t+2.
//     Nothing but the DMA wait and ONE barrier sits between two MFMA loops.
// VMEM order inside tile t: DMA of the next panels, stores of tile t-1, loads of tile t+1.  vmcnt
// retires in order, so `s_waitcnt vmcnt(#stores + #loads)` at the end of the loop waits for exactly
// the DMA (hipcc does not count the asm DMA; its own waits only become a little earlier than needed).
// ----------------------------------------------------------------------------
__device__ static inline void dd_dma16(const double *gsrc, unsigned lds_dst)
{
    unsigned keep;                          // M0 = wave-uniform LDS byte address; lane l lands at +16 l
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ static inline void dd_wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits on gfx9");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// The P tiles are stored WRITE-THROUGH at agent scope (sc1): a kernel ends when its dirty lines have left the L2s (each XCD has its
// own; the next kernel's workgroups read these tiles from other XCDs), and with plain write-back stores most of the 17 MB this kernel
// writes was still in the L2s at its end.  Written through as they are produced, under the MFMA loops: 11.9 -> 10.7 us back to back,
// 35.7 -> 35.0 us per update (A/B in one session; non-temporal stores: 11.5 / 35.2; profiles/r03_downdate_experiments.txt).
// ... and the tiles are READ non-temporally: every tile is read exactly once, by one workgroup -- it need not push the panels (which
// several workgroups share) out of the L2 (34.4 -> 34.1 us per update).
#define DD_LOAD(p) __builtin_nontemporal_load(p)
// (s_nop: the hazard recogniser does not see into the asm -- a VALU write to the data registers of a 128-bit store needs wait states
// behind it, and the registers here are often temporaries the compiler refills at once: without them the tiles came out corrupted)
__device__ static inline void dd_store_sc1(v2d *p, v2d v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
#define DD_STORE(p, v) dd_store_sc1((p), (v))
// LOWER TRIANGLE ONLY (round 3).  The update K (H P) = P H^T S^-1 H P is symmetric and the filter stores P as its lower triangle
// (element (i, j) is valid iff i >= j; the memory above the diagonal is never read by any kernel -- ekf_dev.h): only the tiles on and
// below the diagonal are computed, read and written: half the MFMA work, half the P reads AND half the P writes of the full
// square (round 2 wrote every off-diagonal tile twice, as P(I,J) and transposed as P(J,I): 59.5 MB per launch; now ~43 MB).
// A diagonal tile computes all of its sums, adds them to its P block and then mirrors its own lower half into its upper half
// (an LDS transpose of the RESULT: the values above the diagonal are written for free but never read back), so nothing ever
// depends on two independently rounded halves -- the property that keeps the filter stable.
#ifdef REKF_DEBUG_ENTRY
__device__ long long g_dd_times[1024][2];
extern "C" int rekf_debug_dd_times(long long *out, int n_wg)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dd_times), sizeof(long long) * 2 * (size_t)(n_wg < 1024 ? n_wg : 1024));
}
#endif
// The body is shared by three kernels: k_downdate2 (the downdate alone), k_dd_front (the downdate of scan t enqueued with scan t+1,
// whose front end runs in extra workgroups beside it) -- both with a STATIC tile schedule over their nwg workgroups -- and k_mid (one
// launch per scan, round 5: the downdate of scan t as a role beside scan t+1's mid role, from d.P into d.P_out, its tiles handed out
// by a QUEUE, RekfCtl::dd_queue, to whichever workgroup is free: the pure downdate workgroups from the start, the front role's
// workgroups once their observation is matched).
// No border strips any more (rounds 1-4 treated a border of <= 4 rows past a multiple of 64 as strips on the diagonal tiles; the
// downdate has left the update's critical path, and a last tile row that is mostly padding costs a few tiles of a launch that waits
// for nobody): the tile grid covers roundup(n, 64)^2, rows >= n of the panels are zero.
template <int KC, bool QUEUE>
__device__ __forceinline__ void dd_body(const RekfDev &d, double *dd_smem, int wg, int nwg, unsigned *queue, bool pub_wg)
{
    // dd_smem: [Kn 0 | Kn 1 | HPt 0 | HPt 1] panels, KC x 64 doubles each (>= 33 KiB in all: a diagonal tile's transpose goes through it)
#ifdef REKF_DEBUG_TIMING
    __shared__ long long tq2[24];          // (in LDS: a register array costs every thread 48 VGPRs and changes what is being measured)
    int nq2 = 0;
#ifdef REKF_DEBUG_DD2
#ifndef REKF_DEBUG_DD2_BLOCK
#define REKF_DEBUG_DD2_BLOCK 0
#endif
    const bool rec2 = wg == REKF_DEBUG_DD2_BLOCK && threadIdx.x == 0;
    if (wg == 0 && threadIdx.x == 0) const_cast<RekfCtl *>(d.ctl)->dbg[3] = wall_clock64();   // block 0's entry, for the offset of the recorded block
#else
    const bool rec2 = false;
#endif
    const long long t_entry2 = clock64(), w_entry2 = wall_clock64();
#define D2MARK() do { __builtin_amdgcn_sched_barrier(0); if (rec2 && nq2 < 24) tq2[nq2++] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define D2MARK()
#endif
#ifdef REKF_DEBUG_ENTRY
    // entry / exit wall clock (100 MHz) of EVERY workgroup, release-build register footprint (scripts/gpu_dbg_entry.py)
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_dd_times[blockIdx.x][0] = wall_clock64();
#endif
    // With n known to the host nothing here depends on the control block: k_mid leaves zero panels behind a scan without
    // matches, so the kernel may run unconditionally and its first loads go out one memory round trip earlier.
    const RekfCtl *ctl = d.ctl;
    // the scan's pending Predict (RekfCtl::pred, see there): applied to the tiles of column 0 as they are read, so that what this
    // launch stores there is predicted AND updated.  Fetched here, used after the first MFMA loop at the earliest.
    const bool pred_on = d.pred_slot >= 0;
    __shared__ double s_pred[12];
    double pred_v = 0.0;
    if (pred_on && threadIdx.x >= 64 && threadIdx.x < 64 + 11) pred_v = ((const double *)&ctl->pred[d.pred_ix & 3])[threadIdx.x - 64];   // ab[0], ab[1], C9[0..8]
    // the pose block after this update, as k_mid evaluated and published it (RekfCtl::post_C9): tile (0, 0) stores THOSE bits
    __shared__ double s_post[9];
    __shared__ int s_item;
    const double *post_src = ctl->post_C9[d.post_slot & 1];
    int n = d.n_known;
    if (n < 0) {
        n = ctl->n;
        // (with a Predict pending the kernel runs even so: k_mid has left zero panels, and the tiles of column 0 commit the Predict)
    }
    // The publisher (the scan's last downdate): pose mean, the pose block AFTER this update (RekfCtl::post_C9: k_mid evaluated it; tile
    // (0, 0) below stores the same bits), the n the state will have once the k_augment behind this kernel has run, and the flags --
    // at the START of the kernel: everything is known, and the stores are long through when the kernel ends
    if (d.pub && pub_wg) {
        const int l = threadIdx.x;
        if (l < 3) host_slot_store(d.pub + l, d.mu[l], d.pub_seq, 0);
        else if (l < 12) host_slot_store(d.pub + l, post_src[l - 3], d.pub_seq, 0);
        else if (l == 12) host_slot_store(d.pub + 12, (double)(n + (d.pub_aug ? 2 * ctl->n_new : 0)), d.pub_seq,
                                          __hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    if (d.n_known < 0 && ctl->m == 0 && !pred_on) return;   // nothing matched and nothing pending: P stays as it is
    constexpr int NK = KC / 4;               // MFMA k-steps per tile
    constexpr int ND = KC / 8;               // DMA instructions per panel per wave (2 k-rows each, 4 waves)
    constexpr int PANEL = KC * 64;           // doubles per panel
    static_assert(KC % 16 == 0 && KC >= 16 && KC <= 64, "one k-chunk");
    constexpr int Q4 = NK / 4;               // k-steps per VMEM phase: [DMA Kn][stores (+ DMA HPt)][P loads][-]
    const int T = (n + DT - 1) / DT;
    const size_t ld = (size_t)d.ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int idx = lane & 15, kq = lane >> 4;
    const int wi = wave & 1, wj = wave >> 1;
    const double *__restrict__ Kn = d.Kn;
    const double *__restrict__ HPt = d.HPt;
    const double *__restrict__ P = d.P;
    double *__restrict__ Pout = d.P_out;
    if (pred_on && tid >= 64 && tid < 64 + 11) s_pred[tid - 64] = pred_v;      // (read behind the first item's barriers)

    // LDS: [Kn buffer 0 | Kn buffer 1 | HPt buffer 0 | HPt buffer 1], PANEL doubles each
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)dd_smem;
    auto kn_buf = [&](int b) __attribute__((always_inline)) -> const double * { return dd_smem + (size_t)b * PANEL; };
    auto hp_buf = [&](int b) __attribute__((always_inline)) -> const double * { return dd_smem + (size_t)(2 + b) * PANEL; };
    // one DMA instruction: k-rows 2(4q + wave) + {0,1} of the panel that starts at row `row0` of `src`
    auto dma_piece = [&](const double *src, int row0, int buf_index, int q) __attribute__((always_inline)) {
        const int pr = 4 * q + wave;
        const double *g = src + (size_t)(row0 + 2 * (lane & 31)) + (size_t)(2 * pr + (lane >> 5)) * ld;
        dd_dma16(g, lds0 + (unsigned)(buf_index * PANEL * 8 + pr * 1024));
    };
    auto p_off = [&](int I, int J) __attribute__((always_inline)) -> size_t {
        return (size_t)(DT * I + 32 * wi + 2 * idx) + (size_t)(DT * J + 32 * wj + 2 * kq) * ld;
    };

    // ---- one work item: class A (classA: the tile below the diagonal tile (w, w), then that tile) or a run [t_begin, t_end) of the
    // tiles with I >= J + sub, column by column.  The straight-line forms below need short runs (<= 4 tiles).
    auto run_item = [&](const bool classA, const int w, const int sub, const int t_begin, const int t_end, const bool again) __attribute__((always_inline)) {
    if (t_begin >= t_end) return;
    const int nt = t_end - t_begin;
    // (a class-A workgroup takes its diagonal tile LAST: the mirror then rides on a tile that has nothing to prefetch)
    const int TT = T - sub;                                 // side of the triangle class B enumerates (I >= J + sub)
    // tile number -> (row, column) of the triangle, with a cursor (column, its first tile number) that moves to the queried tile:
    // a workgroup asks for a handful of neighbouring tiles, so after the first query (<= T scalar steps) every look-up is O(1).
    int cur_J = 0, cur_c0 = 0;
    if (TT > 0) {                                           // the cursor starts at a closed-form estimate of the first tile's column (one sqrtf, once)
        const int t0 = t_begin;
        const float bq = 2.0f * (float)TT + 1.0f;
        int Jg = (int)((bq - sqrtf(fmaxf(bq * bq - 8.0f * (float)t0, 0.0f))) * 0.5f);
        Jg = max(0, min(TT - 1, Jg));
        cur_J = Jg; cur_c0 = Jg * TT - (Jg * (Jg - 1)) / 2;
    }
    auto tri_IJ = [&](int tt, int &It, int &Jt) __attribute__((always_inline)) {
        while (cur_J + 1 < TT && tt >= cur_c0 + (TT - cur_J)) { cur_c0 += TT - cur_J; ++cur_J; }
        while (cur_J > 0 && tt < cur_c0) { --cur_J; cur_c0 -= TT - cur_J; }
        Jt = cur_J; It = cur_J + (tt - cur_c0);
    };
    auto tile_IJ = [&](int pos, int &I, int &J) __attribute__((always_inline)) {
        if (classA) { I = (pos == 0 && nt == 2) ? w + 1 : w; J = w; }
        else { tri_IJ(t_begin + pos, I, J); I += sub; }
    };
    // P block of a tile -> P + acc -> store source.  NB register blocks in rotation: the read stream runs AHEAD tiles ahead
    // of the MFMAs.
#ifndef REKF_DD_AHEAD
#define REKF_DD_AHEAD 1
#endif
    constexpr int AHEAD = REKF_DD_AHEAD, NB = AHEAD + 1;
    v2d pq[NB][8];
    v4d acc[2][2];
    int I, J, kb = 0, hb = 0;
    tile_IJ(0, I, J);
    // The tile list of a short range, ONCE, in scalar registers: the straight-line forms below index it with compile-time positions.
    int tI[4] = {I, I, I, I}, tJ[4] = {J, J, J, J};
    if (nt <= 4) {
        if (classA) { tI[1] = w; tI[2] = w; tI[3] = w; }                      // (w + 1, w) then (w, w), or (w, w) alone
        else {
            int ii = I - sub, jj = J;                                         // triangle coordinates: column jj holds ii = jj .. TT - 1
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                if (ii + 1 < TT) ++ii; else { ++jj; ii = jj; }
                tI[q] = ii + sub; tJ[q] = jj;
            }
        }
    }
    // position -> tile: an integral_constant position reads the list, a run-time position (the generic loop) asks the cursor
    auto coords = [&](auto pos_c, int &Iq, int &Jq) __attribute__((always_inline)) {
        if constexpr (std::is_integral<decltype(pos_c)>::value) tile_IJ(pos_c, Iq, Jq);
        else { constexpr int q = decltype(pos_c)::value; static_assert(q >= 0 && q < 4, "short ranges only"); Iq = tI[q]; Jq = tJ[q]; }
    };
    auto shift = [](auto pos_c, auto delta_c) __attribute__((always_inline)) {
        if constexpr (std::is_integral<decltype(pos_c)>::value) return pos_c + decltype(delta_c)::value;
        else return std::integral_constant<int, decltype(pos_c)::value + decltype(delta_c)::value>();
    };

    D2MARK();                                // 0: tile assignment done
    // ---- prologue: both panels of tile 0 by DMA, its P block.  (A further item of a queue-fed workgroup: every wave must be through
    // with the panels -- and the transpose scratch -- of the last one first.)
    if (again) lds_barrier();
    double post_v = 0.0;
    const bool has00 = classA && w == 0;                    // this item ends on tile (0, 0): the pose block by value
    if (has00 && tid >= 128 && tid < 128 + 9) post_v = post_src[tid - 128];
#pragma unroll
    for (int q = 0; q < ND; ++q) dma_piece(Kn, DT * I, 0, q);
#pragma unroll
    for (int q = 0; q < ND; ++q) dma_piece(HPt, DT * J, 2, q);
    {
        const double *Pw = P + p_off(I, J);
#pragma unroll
        for (int q = 0; q < 8; ++q) pq[0][q] = DD_LOAD((const v2d *)(Pw + (size_t)(8 * (q & 3) + (q >> 2)) * ld));
    }
    if (AHEAD > 1 && nt > 1) {               // ... and the P block of tile 1
        int I1, J1;
        tile_IJ(1, I1, J1);
        const double *Pw = P + p_off(I1, J1);
#pragma unroll
        for (int q = 0; q < 8; ++q) pq[NB - 2][q] = DD_LOAD((const v2d *)(Pw + (size_t)(8 * (q & 3) + (q >> 2)) * ld));
        dd_wait_vmcnt<16>();                 // the DMAs (and everything before them); the 16 P loads may still fly
    } else dd_wait_vmcnt<8>();
    if (has00 && tid >= 128 && tid < 128 + 9) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); s_post[tid - 128] = post_v; }
    lds_barrier();
    D2MARK();                                // 1: panels of tile 0 landed

    // ---- one tile.  PAR = which third of pq holds this tile's P block; FIRST (no tile before it: nothing to store), LOAD2
    // (tile pos+2 exists: request its P block), LAST and DIAGSYM are compile-time so that every load and store of a variant
    // is unconditional: hipcc's s_waitcnt pass then counts them exactly (with `if (pos > 0)` around the stores it had to
    // assume the fewest, and waited vmcnt(0) for a P block it had only just requested).
    auto tile_body = [&](auto par_c, auto first_c, auto load2_c, auto last_c, auto special_c, auto pos) __attribute__((always_inline)) {
        using Plus1 = std::integral_constant<int, 1>;
        using PlusA = std::integral_constant<int, AHEAD>;
        using Minus1 = std::integral_constant<int, -1>;
        constexpr int PAR = decltype(par_c)::value, PREV = (PAR + AHEAD) % NB;   // PREV: tile pos-1's block = where tile pos+AHEAD's goes
        constexpr bool FIRST = decltype(first_c)::value, LOAD2 = decltype(load2_c)::value, LAST = decltype(last_c)::value,
                       DIAGSYM = decltype(special_c)::value >= 1;          // diagonal tile: its upper half := mirror of its (new) lower half
        static_assert(!(LAST && LOAD2), "no tile after the last");
        static_assert(!DIAGSYM || LAST, "a diagonal tile ends its workgroup's range (class A)");
        int In = I, Jn = J;
        if constexpr (!LAST) coords(shift(pos, Plus1()), In, Jn);
        const bool needK = !LAST && In != I, needH = !LAST && Jn != J;
        const double *Pn = nullptr;                         // tile pos+2's P block
        if constexpr (LOAD2) { int I2, J2; coords(shift(pos, PlusA()), I2, J2); Pn = P + p_off(I2, J2); }
        double *Po = nullptr;                               // where tile pos-1 goes
        if constexpr (!FIRST) { int Ip, Jp; coords(shift(pos, Minus1()), Ip, Jp); Po = Pout + p_off(Ip, Jp); }
        const double *aW = hp_buf(hb) + 32 * wj + 2 * idx + kq * 64;        // A[j][k] = HP(k,j)
        const double *bK = kn_buf(kb) + 32 * wi + 2 * idx + kq * 64;        // B[k][i] = Kn(i,k)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0, 0, 0, 0};
        v2d a2 = *(const v2d *)(aW), b2 = *(const v2d *)(bK);
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            v2d a2n = a2, b2n = b2;
            if (kk + 1 < NK) {
                a2n = *(const v2d *)(aW + (kk + 1) * 256);
                b2n = *(const v2d *)(bK + (kk + 1) * 256);
            }
            // the operand reads of step kk+1 are ISSUED before this step's MFMAs: without the fence hipcc gives a2n / b2n the registers
            // of a2 / b2 and sinks the ds_reads below the MFMAs that still read them (seen in the ISA, round 3)
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.x, b2.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.x, b2.y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.y, b2.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.y, b2.y, acc[1][1], 0, 0, 0);
            // ---- this k-step's share of the VMEM traffic (compile-time positions)
            // phase 0: DMA Kn; 1: the first half of the previous tile's stores, then DMA HPt; 2: the second half; 3: P loads.
            // The DMA goes FIRST: the wait at the end of the loop is for the oldest operations only, so the stores behind the DMA
            // stay in flight across the tile boundary instead of having to be acknowledged inside it (all eight stores in one
            // phase stall the MFMAs queued behind them on the CU's store path)
            const int ph = kk / Q4, off = kk % Q4;
            constexpr int PH_LOAD = 3;
            if (ph == 0) {
                if (!LAST && needK) {
#pragma unroll
                    for (int q = 0; q < ND; ++q)
                        if ((q * Q4) / ND == off) dma_piece(Kn, DT * In, kb ^ 1, q);
                }
            } else if (ph == 1) {
                if (!FIRST) {
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                        if ((x * Q4) / 4 == off) DD_STORE((v2d *)(Po + (size_t)(8 * (x & 3) + (x >> 2)) * ld), pq[PREV][x]);
                }
                if (!LAST && needH) {
#pragma unroll
                    for (int q = 0; q < ND; ++q)
                        if ((q * Q4) / ND == off) dma_piece(HPt, DT * Jn, 2 + (hb ^ 1), q);
                }
            } else if (ph == 2) {
                if (!FIRST) {
#pragma unroll
                    for (int x = 4; x < 8; ++x)
                        if (((x - 4) * Q4) / 4 == off) DD_STORE((v2d *)(Po + (size_t)(8 * (x & 3) + (x >> 2)) * ld), pq[PREV][x]);
                }
            }
            if (ph == PH_LOAD && LOAD2) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if ((q * Q4) / 8 == off) pq[PREV][q] = DD_LOAD((const v2d *)(Pn + (size_t)(8 * (q & 3) + (q >> 2)) * ld));
            }
            __builtin_amdgcn_sched_barrier(0);
            a2 = a2n; b2 = b2n;
        }
        D2MARK();                            // MFMA loop done
        if (pred_on && J == 0) {
            // the pending Predict on the tile as read: columns 0, 1 against column 2 (register 0 of the lanes 16 up: kq = 1), the same
            // single operations the covariance pass of the front kernel used to do in memory; the 3 x 3 pose block by value
#pragma clang fp contract(off)
            const double pa = s_pred[0], pb = s_pred[1];
            const double c2x = __shfl_down(pq[PAR][0].x, 16, 64), c2y = __shfl_down(pq[PAR][0].y, 16, 64);
            if (wj == 0 && kq == 0) {
                const int r0 = DT * I + 32 * wi + 2 * idx;              // this lane's rows r0, r0 + 1; register 0 = column 0, register 4 = column 1
                if (r0 >= 4) {
                    pq[PAR][0].x = pq[PAR][0].x + pa * c2x; pq[PAR][4].x = pq[PAR][4].x + pb * c2x;
                    pq[PAR][0].y = pq[PAR][0].y + pa * c2y; pq[PAR][4].y = pq[PAR][4].y + pb * c2y;
                } else if (r0 == 2) {                                    // row 3 is an ordinary row; rows 0..2 are the pose block, which
                    pq[PAR][0].y = pq[PAR][0].y + pa * c2y; pq[PAR][4].y = pq[PAR][4].y + pb * c2y;     // tile (0, 0) takes by value below
                }
            }
        }
        // P + sum_k (the P block was requested a tile ago)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pq[PAR][mt * 4 + r].x += acc[mt][0][r];
                pq[PAR][mt * 4 + r].y += acc[mt][1][r];
            }
        if (DIAGSYM && I == 0 && wave == 0 && idx < 2 && kq < 2) {
            // the pose block: the values k_mid evaluated and published (lower triangle; the mirror below fills the rest)
            if (idx == 0 && kq == 0) { pq[PAR][0].x = s_post[0]; pq[PAR][0].y = s_post[1]; pq[PAR][4].y = s_post[4]; }
            if (idx == 1 && kq == 0) { pq[PAR][0].x = s_post[2]; pq[PAR][4].x = s_post[5]; }
            if (idx == 1 && kq == 1) pq[PAR][0].x = s_post[8];
        }
        if (DIAGSYM) {
            // the NEW tile through LDS, S[j][i] = element (i, j) (row stride 66: the lanes of a 16-group differ in i); an element
            // above the diagonal (i < j) then takes S[i][j] = the new element (j, i): the upper half is the mirror image of the
            // lower half whatever the memory above the diagonal held.  Every panel is dead by now (a diagonal tile is the last of
            // its item): S takes the front of the LDS arena (33 KiB; the smallest launch has 32 KiB of panels + scratch).
            lds_barrier();
            double *S = dd_smem;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int jj = 32 * wj + 2 * kq + 8 * r + mt, ii = 32 * wi + 2 * idx;
                    S[jj * 66 + ii] = pq[PAR][mt * 4 + r].x;
                    S[jj * 66 + ii + 1] = pq[PAR][mt * 4 + r].y;
                }
            lds_barrier();
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int jj = 32 * wj + 2 * kq + 8 * r + mt, ii = 32 * wi + 2 * idx;
                    const int dij = ii - jj;                                          // i - j for the .x element
                    if (dij < 0) pq[PAR][mt * 4 + r].x = S[ii * 66 + jj];
                    if (dij + 1 < 0) pq[PAR][mt * 4 + r].y = S[(ii + 1) * 66 + jj];
                }
        }
        D2MARK();                            // P block there, combined
        if (LAST) {
            double *Pw = Pout + p_off(I, J);
#pragma unroll
            for (int q = 0; q < 8; ++q) DD_STORE((v2d *)(Pw + (size_t)(8 * (q & 3) + (q >> 2)) * ld), pq[PAR][q]);
            return;
        }
        if (needH) dd_wait_vmcnt<(!FIRST ? 4 : 0) + (LOAD2 ? 8 : 0)>();             // after the last HPt DMA: (the second half of the stores +) this tile's 8 P loads
        else if (needK) dd_wait_vmcnt<(FIRST ? 0 : 8) + (LOAD2 ? 8 : 0)>();          // after the last Kn DMA: (all of the previous tile's stores +) (8 P loads)
        if (needK || needH) lds_barrier();
        if (needK) kb ^= 1;
        if (needH) hb ^= 1;
        I = In; J = Jn;
        D2MARK();                            // next panels landed, barrier passed
    };
    using Tt = std::true_type;
    using Ff = std::false_type;
    using C0 = std::integral_constant<int, 0>;      // tile kinds: off-diagonal / diagonal
    using C1 = std::integral_constant<int, 1>;
    // Short ranges (the BASELINE sizes) run as STRAIGHT-LINE code, one instantiation per position: with no loop and no join in the
    // way, hipcc's s_waitcnt pass places every wait exactly (through the generic loop below it merges the variants' states at the
    // joins and waits for far younger loads than the P block it needs).  A diagonal tile ends its (class A) range, so only the last
    // position has the diagonal variant.  The two classes take separate code.
    auto straight = [&](auto nt_c, auto cls_a) __attribute__((always_inline)) {
        constexpr int NT = decltype(nt_c)::value;
        constexpr bool CLS_A = decltype(cls_a)::value;
        auto one = [&](auto pos_c) __attribute__((always_inline)) {
            constexpr int POS = decltype(pos_c)::value;
            using Par = std::integral_constant<int, POS % NB>;
            using First = std::integral_constant<bool, POS == 0>;
            using Load2 = std::integral_constant<bool, (POS + AHEAD < NT)>;
            using Last = std::integral_constant<bool, POS == NT - 1>;
            if constexpr (POS == NT - 1 && CLS_A) tile_body(Par(), First(), Load2(), Last(), C1(), pos_c);
            else tile_body(Par(), First(), Load2(), Last(), C0(), pos_c);
        };
        one(std::integral_constant<int, 0>());
        if constexpr (NT > 1) one(std::integral_constant<int, 1>());
        if constexpr (NT > 2) one(std::integral_constant<int, 2>());
        if constexpr (NT > 3) one(std::integral_constant<int, 3>());
    };
    if (classA) {
        if (nt == 2) straight(std::integral_constant<int, 2>(), Tt());
        else straight(std::integral_constant<int, 1>(), Tt());
    } else if (nt <= 4) {
        if (nt == 3) straight(std::integral_constant<int, 3>(), Ff());
        else if (nt == 2) straight(std::integral_constant<int, 2>(), Ff());
        else if (nt == 4) straight(std::integral_constant<int, 4>(), Ff());
        else straight(std::integral_constant<int, 1>(), Ff());
    } else {
        auto run4 = [&](auto par_c, auto first_c, auto load2_c, auto last_c, int pos) __attribute__((always_inline)) {
            tile_body(par_c, first_c, load2_c, last_c, C0(), pos);
        };
        auto run = [&](auto par_c, int pos) __attribute__((always_inline)) {               // a tile after the first
            if (pos == nt - 1) run4(par_c, Ff(), Ff(), Tt(), pos);
            else if (pos + AHEAD < nt) run4(par_c, Ff(), Tt(), Ff(), pos);
            else run4(par_c, Ff(), Ff(), Ff(), pos);
        };
        using B0 = std::integral_constant<int, 0>;
        using B1 = std::integral_constant<int, 1>;
        using B2 = std::integral_constant<int, NB - 1>;
        if (nt <= AHEAD) run4(B0(), Tt(), Ff(), Ff(), 0);
        else run4(B0(), Tt(), Tt(), Ff(), 0);
        for (int pos = 1; pos < nt; pos += NB) {            // pq blocks by name: NB tiles per trip
            run(B1(), pos);
            if (NB == 3 && pos + 1 < nt) run(B2(), pos + 1);
            if (pos + NB - 1 < nt) run(B0(), pos + NB - 1);
        }
    }
    };      // run_item

    if constexpr (QUEUE) {
        // ---- tiles from the queue: items 0 .. T-1 = class A (diagonal tile w behind the tile below it), then runs of DD_RUN tiles of
        // the triangle I >= J + 2, column by column
        constexpr int DD_RUN = 3;
        const int nB = (T - 1) * (T - 2) / 2, n_items = T + (nB + DD_RUN - 1) / DD_RUN;
        bool again = false;
        for (;;) {
            if (tid == 0) s_item = (int)atomicAdd(queue, 1u);
            __syncthreads();
            const int it = __builtin_amdgcn_readfirstlane(s_item);
            __syncthreads();
            if (it >= n_items) break;
            const bool cla = it < T;
            run_item(cla, cla ? it : 0, 2, cla ? 0 : DD_RUN * (it - T), cla ? ((it + 1 < T) ? 2 : 1) : min(DD_RUN * (it - T) + DD_RUN, nB), again);
            again = true;
        }
    } else {
        int w = wg;
        const int nw = nwg;
        // the triangle column by column (tile column J: I = J .. T-1); an XCD's workgroups take consecutive ranges of it
        if (nw >= 8 && (nw & 7) == 0) w = (wg & 7) * (nw >> 3) + (wg >> 3);
        // Two classes of workgroups.  Class A, workgroups [0, T): the diagonal tile (w, w) -- it costs more than an ordinary tile: the
        // symmetric finish -- behind the tile below it, (w+1, w), which shares its HPt panel.  Class B, the rest: the tiles with
        // I >= J + 2, column by column, in equal ranges.  (When class B gets fewer than three tiles per workgroup -- small states --
        // class A keeps to its diagonal tile and the tiles below the diagonal join class B: dd_sub = 1.)
        const bool classA = w < T;
        // class B: every free workgroup takes tiles -- lo each, the first x of them one more.  (lo, x) come from the host when it knows n
        // exactly (no division in the prologue), else they are derived here from the real T and the grid the host sized by its
        // bound of n (downdate_schedule below, same arithmetic)
        int sub = (d.dd_sub == 1) ? 1 : 2, lo = d.dd_lo, xhi = d.dd_x;
        if (d.dd_sub == 0) {
            const unsigned room = (unsigned)((nw - T > 1) ? nw - T : 1);
            unsigned nBq = (unsigned)((T - 1) * (T - 2) / 2);
            sub = 2;
            if ((nBq + room - 1) / room < 3) { sub = 1; nBq = (unsigned)(T * (T - 1) / 2); }
            lo = (int)(nBq / room); xhi = (int)(nBq - (unsigned)lo * room);
        }
        const int wq = w - T;
        const int nB = (T - sub + 1) * (T - sub) / 2;                  // class B: its tiles (the ranges are clamped to them whatever the host planned)
        const int t_begin = classA ? 0 : min(wq * lo + min(wq, xhi), nB);
        const int t_end = classA ? ((sub == 2 && w + 1 < T) ? 2 : 1) : min(wq * lo + min(wq, xhi) + lo + (wq < xhi ? 1 : 0), nB);
        run_item(classA, w, sub, t_begin, t_end, false);
    }
#ifdef REKF_DEBUG_ENTRY
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (this wave's last stores acknowledged)
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_dd_times[blockIdx.x][1] = wall_clock64();
#endif
#ifdef REKF_DEBUG_TIMING
    if (rec2) {
        RekfCtl *c = const_cast<RekfCtl *>(d.ctl);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        c->dbg[6] = clock64() - t_entry2;               // whole body, last stores retired
        c->dbg[5] = wall_clock64() - w_entry2;          // same in 100 MHz ticks
        c->dbg[7] = nq2;
        c->dbg[4] = w_entry2;                           // this block's entry (100 MHz wall clock)
        for (int i = 0; i < nq2; ++i) c->dbg[8 + i] = tq2[i] - t_entry2;
    }
#endif
}

template <int KC>
__global__ __launch_bounds__(256) void k_downdate2(RekfDev d)
{
    extern __shared__ __attribute__((aligned(16))) double dd_smem_k[];
    dd_body<KC, false>(d, dd_smem_k, (int)blockIdx.x, (int)gridDim.x, nullptr, blockIdx.x == 0);
}
template <int KC>
__global__ __launch_bounds__(256) void k_dd_front(RekfDev d, RekfDev dn, RekfFrontArgs A)
{
    if ((int)blockIdx.x >= d.dd_grid) {               // the front end of the NEXT scan: workgroups of their own, nothing of P read or written
#ifdef REKF_DEBUG_ENTRY
        if (threadIdx.x == 0 && blockIdx.x < 1024) g_dd_times[blockIdx.x][0] = wall_clock64();
#endif
        front_role<256>(dn, A, (int)blockIdx.x - d.dd_grid, (int)gridDim.x - d.dd_grid, true);
#ifdef REKF_DEBUG_ENTRY
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0 && blockIdx.x < 1024) g_dd_times[blockIdx.x][1] = wall_clock64();
#endif
        return;
    }
    extern __shared__ __attribute__((aligned(16))) double dd_smem_k[];
    dd_body<KC, false>(d, dd_smem_k, (int)blockIdx.x, d.dd_grid, nullptr, blockIdx.x == 0);
    if (d.aug_tail) {
        // The scan this downdate belongs to may have met new reflectors (a filter below its capacity): their covariance rows (cc:311-364) are
        // functions of the DOWNDATED pose columns, i.e. of what this launch's workgroups are writing.  No launch of its own (k_augment:
        // a kernel boundary and an early-out per scan) and no waiting workgroup: when there IS something to append -- every workgroup
        // reads the same n2 -- each one releases its tiles and counts, and the one that counts last appends.
        RekfCtl *ctl = d.ctl;
        const RekfCtl::AugRec *ar = &ctl->augrec[(d.aug_tail - 1) & 1];
        const int n2 = ar->n2;
        if (n2 > 0) {
            __shared__ int s_last_dd;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            if (threadIdx.x == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                s_last_dd = __hip_atomic_fetch_add(&ctl->aug_arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == d.dd_grid - 1;
            }
            __syncthreads();
            if (s_last_dd) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                const int nb = ar->n_before;
                double *scr = dd_smem_k;                 // (the downdate's panels are dead)
                augment_rows(d, nb, n2, A.obs_cov, (double (*)[6])scr, scr + 6 * REKF_MAX_OBS_DEV, scr + 6 * REKF_MAX_OBS_DEV + 9, 256,
                             [&](int k, float &rx, float &ry) { rx = ar->obs[2 * k]; ry = ar->obs[2 * k + 1]; });
                if (threadIdx.x == 0) { ctl->n = nb + 2 * n2; ctl->aug_arrive = 0; }      // cc:360-363
            }
        }
    }
}

// ----------------------------------------------------------------------------
// k_augment (cc:311-364): one workgroup; runs only while the map is growing.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_augment(RekfDev d, RekfFrontArgs A)
{
    __shared__ double Gp[REKF_MAX_OBS_WIDE][6];
    __shared__ double Sxi[9];
    __shared__ double RQR[4];
    RekfCtl *ctl = d.ctl;
    const int N2 = ctl->n_new;
    const int n = ctl->n;
    if (N2 == 0) return;
    augment_rows(d, n, N2, A.obs_cov, Gp, Sxi, RQR, 256, [&](int k, float &rx, float &ry) {
        const int local_id = ctl->new_ids[k];
        rx = rekf_obs(A, 2 * local_id); ry = rekf_obs(A, 2 * local_id + 1);
    });
    if (threadIdx.x == 0) ctl->n = n + 2 * N2;                      // cc:360-363
}

// GetState's pose part (ekf_slam.h GetState / ros_node.cc's pose publisher): mu[0..2], the 3 x 3 pose block, n and the error
// flags, straight into the host's slots
__global__ void k_publish_pose(RekfDev d, RekfHostSlot *out, int seq)
{
    const int l = threadIdx.x;
    if (l < 3) host_slot_store(out + l, d.mu[l], seq, 0);
    else if (l < 12) host_slot_store(out + l, rekf_plower(d.P, d.ld, (l - 3) % 3, (l - 3) / 3), seq, 0);
    else if (l == 12) host_slot_store(out + 12, (double)d.ctl->n, seq, d.ctl->err);
}

// ----------------------------------------------------------------------------
// PredictState, the O(n) part of what a predict changes (cc:97-152), non-mutating: the predicted rows 0,1 and columns 0,1 of
// P (Predict's own arithmetic, elementwise).  The pose and the 3x3 pose block come from the host's mirror.
// out: row0[ld] | row1[ld] | col0[ld] | col1[ld]
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_predict_rows(RekfDev d, RekfFrontArgs A, double *out)
{
#pragma clang fp contract(off)
    // (the motion model runs on the host's pose mirror, which also supplies the predicted pose and pose block: A.pre_ab = this
    // predict's (a, b); the arithmetic on the rows is k_apply_predict's, elementwise, so a later mutating Predict gives these bits)
    const int n = d.ctl->n;
    const size_t ld = (size_t)d.ld;
    const double *P = d.P;
    const double a = A.pre_ab[0], b = A.pre_ab[1];
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
        if (idx < 3) continue;
        const double p2 = P[idx + 2 * ld];
        const double n0 = P[idx + 0 * ld] + a * p2, n1 = P[idx + 1 * ld] + b * p2;
        out[2 * ld + idx] = n0;            // column 0
        out[3 * ld + idx] = n1;            // column 1
        out[0 * ld + idx] = n0;            // row 0 (P is exactly symmetric)
        out[1 * ld + idx] = n1;            // row 1
    }
}
void rekf_launch_predict_rows(const RekfDev &d, const RekfFrontArgs &a, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_predict_rows, dim3(16), dim3(256), 0, s, d, a, out);
}

// ----------------------------------------------------------------------------
// launch wrappers
// ----------------------------------------------------------------------------
void rekf_launch_apply_predict(const RekfDev &d, const RekfFrontArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(k_apply_predict, dim3(1), dim3(1024), 0, s, d, a);
}
void rekf_launch_front_mb(const RekfDev &d, const RekfFrontArgs &a, int n_ub, hipStream_t s)
{
    (void)n_ub;
    hipLaunchKernelGGL(k_front_mb, dim3(FRONT_MB), dim3(1024), 0, s, d, a);
}
void rekf_launch_compact_wide(const RekfDev &d, const RekfFrontArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(k_compact_wide, dim3(1), dim3(REKF_MAX_OBS_WIDE), 0, s, d, a);
}
// per-DEVICE facts the launches below need: the CU count, and which kernel variants have their opt-in to more than 64 KiB of dynamic LDS
constexpr int DD_MAX_DEV = 64;
static struct {
    int n_cu_of[DD_MAX_DEV] = {0};
    unsigned attr_done[DD_MAX_DEV] = {0};     // bit KC/16 : k_downdate2<KC> / k_dd_front<KC>
    unsigned attr_mid[DD_MAX_DEV] = {0};      // bit 3 (NBR / 2 - 1) + MODE : k_mid<NBR, MODE>
    std::mutex mu;
} g_dd_cache;
static int dd_cache_slot(int &dev)            // (call with g_dd_cache.mu held)
{
    (void)hipGetDevice(&dev);
    const int slot = (dev >= 0 && dev < DD_MAX_DEV) ? dev : 0;
    if (g_dd_cache.n_cu_of[slot] == 0 || dev != slot) {
        hipDeviceProp_t prop;
        int cu = 0;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cu = prop.multiProcessorCount;
        if (cu <= 0) cu = 256;
        g_dd_cache.n_cu_of[slot] = cu;
        g_dd_cache.attr_done[slot] = 0;
        g_dd_cache.attr_mid[slot] = 0;
    }
    return slot;
}
// k_mid<NBR, MODE> with `bytes` of dynamic LDS (all of its LDS is one arena: MidLds, and the downdate role's panels over it)
template <int NBR, int MODE> static void launch_mid_as(int grid, int with_dd, hipStream_t s, const RekfDev &d, const RekfFrontArgs &a, const RekfDev &dp, const RekfFrontArgs &an)
{
    constexpr int MID_BYTES = (int)sizeof(MidLds<NBR>);
    constexpr int BYTES = (MODE != 1 && MID_BYTES < REKF_DD_LDS_BYTES) ? REKF_DD_LDS_BYTES : MID_BYTES;     // (one size per kernel: the opt-in is per function)
    static_assert(BYTES <= 160 * 1024 - 12 * 1024, "the front role's and the downdate role's static LDS ride on top");
    (void)with_dd;
    {
        std::lock_guard<std::mutex> guard(g_dd_cache.mu);
        int dev = 0;
        const int slot = dd_cache_slot(dev);
        const unsigned bit = 1u << (4 * (NBR / 2 - 1) + MODE);
        if (!(g_dd_cache.attr_mid[slot] & bit) || dev != slot) {
            (void)hipFuncSetAttribute((const void *)k_mid<NBR, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
            g_dd_cache.attr_mid[slot] |= bit;
        }
    }
    // (the fields the kernel's first instructions need, as leading scalars: k_mid)
    if (a.dd_first > 0xfff || a.n_mid > 0xfff || a.spec_front > 0xff || a.dd_in_mid > 0xfff || a.front_in_mid > 0xff) {
        std::fprintf(stderr, "rekf: k_mid launch header overflow (n_mid %d, dd_first %d, dd_in_mid %d): launch dropped\n", a.n_mid, a.dd_first, a.dd_in_mid);
        return;
    }
    const int h0 = (a.pred_slot & 1) | ((a.pred_ix & 3) << 1) | ((a.corr ? 1 : 0) << 3) | ((a.corr_post & 1) << 4) | ((a.corr_pred_ix & 3) << 5) |
                   ((a.spec ? 1 : 0) << 7) | ((a.corr && a.corr_pred >= 0 ? 1 : 0) << 8) | ((a.dd_par & 1) << 9);
    const int h4 = (a.dd_first & 0xfff) | ((a.n_mid & 0xfff) << 12) | ((a.spec_front & 0xff) << 24);
    const int h5 = (a.dd_in_mid & 0xfff) | ((a.front_in_mid & 0xff) << 12);
    hipLaunchKernelGGL((k_mid<NBR, MODE>), dim3(grid), dim3(512), BYTES, s, d.ctl, h0, a.K, (int)a.scan_id, (int)a.corr_scan, h4, h5, d, a, dp, an);
}
void rekf_launch_mid(const RekfDev &d, RekfFrontArgs &a, int n_ub, int m_ub, bool mode_grow, hipStream_t s)
{
    // m_ub <= 64 (the host checks): one workgroup per 16 state rows
    a.n_mid = (n_ub + MID_ROWS - 1) / MID_ROWS;
    a.dd_in_mid = 0; a.dd_first = 0; a.spec_front = 0;
    const int grid = a.n_mid + (a.front_in_mid > 0 ? a.front_in_mid : 0);
    const int mode = (a.front_in_mid > 0) ? 2 : (mode_grow ? 1 : (a.grid_match ? 3 : 0));
    if (m_ub <= 32) {
        if (mode == 0) launch_mid_as<2, 0>(grid, 0, s, d, a, d, a);          // (the last argument: the downdate role's view, one launch per scan only)
        else if (mode == 1) launch_mid_as<2, 1>(grid, 0, s, d, a, d, a);
        else if (mode == 3) launch_mid_as<2, 3>(grid, 0, s, d, a, d, a);
        else launch_mid_as<2, 2>(grid, 0, s, d, a, d, a);
    } else {
        if (mode == 0) launch_mid_as<4, 0>(grid, 0, s, d, a, d, a);
        else if (mode == 1) launch_mid_as<4, 1>(grid, 0, s, d, a, d, a);
        else if (mode == 3) launch_mid_as<4, 3>(grid, 0, s, d, a, d, a);
        else launch_mid_as<4, 2>(grid, 0, s, d, a, d, a);
    }
}
// ONE launch per scan (round 5): [the scan's front end, front_wgs workgroups (0: it ran before this launch) | its mid role, which takes the
// pending downdate dd as a correction of what it gathers | dd itself, from dd.P into dd.P_out, on the CUs the others leave free, tiles from
// RekfCtl::dd_queue].  The caller has set a.corr / corr_pred / corr_post / dd_par.  Returns the launch's grid.
// The one-launch form needs CUs for three roles at once (one workgroup per CU: the arena).  With fewer than REKF_SCAN_MIN_DD_WGS left for the
// downdate role -- n beyond ~3000 on 256 CUs -- the whole O(n^2 m) downdate would crawl on a handful of workgroups: the host then keeps the
// two-launch chain (k_dd_front with every CU, then k_mid).  Also guards the 12-bit fields of k_mid's launch header (launch_mid_as).
#define REKF_SCAN_MIN_DD_WGS 48
static int device_cu_count()
{
    std::lock_guard<std::mutex> guard(g_dd_cache.mu);
    int dev = 0;
    return g_dd_cache.n_cu_of[dd_cache_slot(dev)];
}
bool rekf_scan_launch_fits(int n_ub, int K_front)
{
    const int n_mid = (n_ub + MID_ROWS - 1) / MID_ROWS;
    const int dd_first = (n_mid + (K_front > 0 ? K_front : 0) + 7) & ~7;
    return dd_first + REKF_SCAN_MIN_DD_WGS <= device_cu_count() && dd_first < 0x1000;
}
int rekf_launch_scan(const RekfDev &dd, const RekfDev &d, RekfFrontArgs &a, int n_ub, int m_ub, int front_wgs, const RekfFrontArgs *an, hipStream_t s)
{
    const int n_cu = device_cu_count();
    a.n_mid = (n_ub + MID_ROWS - 1) / MID_ROWS;
    a.front_in_mid = front_wgs;
    a.spec_front = (an && front_wgs == 0) ? an->K : 0;    // the NEXT scan's speculative front end: workgroups behind the mid role's
    a.dd_first = (front_wgs + a.n_mid + a.spec_front + 7) & ~7;          // (the queue does not care; a multiple of 8 keeps block number mod 8 = XCD for everybody)
    const int n_dd = (dd.n_known >= 0) ? dd.n_known : dd.n_max;
    const int T = (n_dd + DT - 1) / DT, nB = (T - 1) * (T - 2) / 2, items = T + (nB + 2) / 3;
    int wgs = n_cu - a.dd_first;                          // one workgroup per CU (the arena), everybody resident from the start
    if (wgs < 8) wgs = 8;                                 // (never on the host's path: rekf_scan_launch_fits)
    if (wgs > items) wgs = items;
    a.dd_in_mid = wgs;
    const int grid = a.dd_first + wgs;
    const int mode = front_wgs > 0 ? 2 : (a.grid_match ? 3 : 0);
    if (m_ub <= 32) {
        if (mode == 0) launch_mid_as<2, 0>(grid, 1, s, d, a, dd, an ? *an : a);
        else if (mode == 3) launch_mid_as<2, 3>(grid, 1, s, d, a, dd, an ? *an : a);
        else launch_mid_as<2, 2>(grid, 1, s, d, a, dd, a);
    } else {
        if (mode == 0) launch_mid_as<4, 0>(grid, 1, s, d, a, dd, an ? *an : a);
        else if (mode == 3) launch_mid_as<4, 3>(grid, 1, s, d, a, dd, an ? *an : a);
        else launch_mid_as<4, 2>(grid, 1, s, d, a, dd, a);
    }
    return grid;
}
template <int KC> static void launch_downdate2(const RekfDev &d, int grid, hipStream_t s, bool first_on_device, const RekfDev *dn, const RekfFrontArgs *an, int n_front)
{
    constexpr int BYTES = 4 * KC * 64 * (int)sizeof(double) + (KC < 64 ? 16384 : 0);      // (a diagonal tile's transpose needs 33 KiB)
    if (first_on_device) {
        (void)hipFuncSetAttribute((const void *)k_downdate2<KC>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
        (void)hipFuncSetAttribute((const void *)k_dd_front<KC>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
    }
    if (dn) hipLaunchKernelGGL((k_dd_front<KC>), dim3(grid + n_front), dim3(256), BYTES, s, d, *dn, *an);
    else hipLaunchKernelGGL((k_downdate2<KC>), dim3(grid), dim3(256), BYTES, s, d);
}
// host half of the STATIC tile schedule (tests/test_downdate_schedule_cpu.py restates it): T class-A workgroups (diagonal tile + the
// one below) + equal ranges of the rest
static void downdate_schedule(int n_ub, int slots, int &grid, int &dd_lo, int &dd_x, int &dd_sub)
{
    const int T = (n_ub + DT - 1) / DT;
    const int room = (slots - T > 1) ? slots - T : 1;
    int nB = (T - 1) * (T - 2) / 2;
    dd_sub = 2;
    if ((nB + room - 1) / room < 3) { dd_sub = 1; nB = T * (T - 1) / 2; }   // small states: a class-A workgroup keeps to its diagonal tile
    dd_lo = nB / room; dd_x = nB - dd_lo * room;          // the first dd_x class-B workgroups take dd_lo + 1 tiles, the others dd_lo
    grid = T + (dd_lo > 0 ? room : dd_x);
    if (grid >= 64) grid = (grid + 7) & ~7;         // multiple of 8 for the per-XCD numbering (workgroups past the last range return at once)
}
// dn / an non-null: the fused form k_dd_front -- scan t's downdate with the front end of scan t+1 (dn, an) in FRONT_MB further workgroups,
// which take their CUs out of the downdate's schedule
static void launch_downdate_any(const RekfDev &d, int n_ub, hipStream_t s, const RekfDev *dn, const RekfFrontArgs *an)
{
    // persistent: one workgroup per CU (its panels fill most of the LDS), never more workgroups than tiles.  The opt-in to
    // more than 64 KiB of dynamic LDS and the CU count are per DEVICE: a process may hold handles on several GPUs.
    // handles are independent (multi-session servers drive them from several host threads) but this cache is per process: the lock
    // is held across the launch, so that no thread launches a variant before the thread that first needed it has opted it in
    std::lock_guard<std::mutex> guard(g_dd_cache.mu);
    int dev = 0;
    const int slot = dd_cache_slot(dev);
    int *const n_cu_of = g_dd_cache.n_cu_of;
    unsigned *const attr_done = g_dd_cache.attr_done;
    int grid, dd_lo, dd_x, dd_sub;
    int n_front = 0;
    if (dn) { n_front = an->K < FRONT_MB ? (an->K > 0 ? an->K : 1) : FRONT_MB; }
    int slots = n_cu_of[slot] * DD_WG_PER_CU - n_front;
    if (slots < 8) slots = 8;
    downdate_schedule(n_ub, slots, grid, dd_lo, dd_x, dd_sub);
    const int kc = (d.kc_ub < 16) ? 16 : ((d.kc_ub > 64) ? 64 : d.kc_ub);    // one k-chunk: the host never asks for more than 64 rows per step
    const unsigned bit = 1u << (kc / 16);
    const bool first = !(attr_done[slot] & bit) || dev != slot;
    attr_done[slot] |= bit;
    RekfDev dp = d;
    dp.dd_lo = dd_lo; dp.dd_x = dd_x; dp.dd_sub = dd_sub;                 // (dp.pred_slot: the caller's)
    dp.dd_grid = dn ? grid : 0;
    if (d.n_known < 0) { dp.dd_lo = 0; dp.dd_x = 0; dp.dd_sub = 0; }   // n_ub is only a bound: the kernel derives the schedule from the real n
    if (kc == 64) launch_downdate2<64>(dp, grid, s, first, dn, an, n_front);
    else if (kc == 48) launch_downdate2<48>(dp, grid, s, first, dn, an, n_front);
    else if (kc == 32) launch_downdate2<32>(dp, grid, s, first, dn, an, n_front);
    else launch_downdate2<16>(dp, grid, s, first, dn, an, n_front);
}
void rekf_launch_downdate(const RekfDev &d, int n_ub, hipStream_t s) { launch_downdate_any(d, n_ub, s, nullptr, nullptr); }
void rekf_launch_dd_front(const RekfDev &d, int n_ub, const RekfDev &dn, const RekfFrontArgs &an, hipStream_t s) { launch_downdate_any(d, n_ub, s, &dn, &an); }
void rekf_launch_augment(const RekfDev &d, const RekfFrontArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(k_augment, dim3(1), dim3(256), 0, s, d, a);
}
// ----------------------------------------------------------------------------
// k_ellipses: the caller's marker ellipses (src/ros_node.cc:750-765), one thread per landmark.
// Eigen-decomposition of the (unsymmetrised) 2x2 block T = [[a,b],[c,d]] in the order/sign convention
// of Eigen 3.3's RealSchur for a 2x2 real matrix (what Eigen::EigenSolver runs, :759-761): if the
// sub-diagonal is negligible, |c| <= eps(|a|+|d|), T is left alone: D = (a,d), V(:,0) = e0; else the
// two-real-roots step of hqr2: p = (a-d)/2, z = sqrt|p^2 + c b|, pz = p +- z (sign of p), D0 = d + pz,
// D1 = d - c b / pz, V(:,0) = (pz, c)/|.|.  So the eigenvalue that stays with T(0,0) comes first.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ellipses(RekfDev d, double *out5, int cap)
{
#pragma clang fp contract(off)
    const int n = d.ctl->n;
    const int L = (n - 3) / 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L || i >= cap) return;
    const size_t ld = (size_t)d.ld;
    const int id = 3 + 2 * i;
    const double a = d.P[id + id * ld], c = d.P[(id + 1) + id * ld], dd = d.P[(id + 1) + (id + 1) * ld];
    const double b = c;                               // sigma(id, id+1): P is stored as its lower triangle
    double d0, d1, vx, vy;
    if (fabs(c) <= 2.220446049250313e-16 * (fabs(a) + fabs(dd))) {
        d0 = a; d1 = dd; vx = 1.0; vy = 0.0;
    } else {
        const double p = 0.5 * (a - dd);
        const double q = p * p + c * b;
        const double z = sqrt(fabs(q));
        const double pz = (p >= 0.0) ? p + z : p - z;
        d0 = dd + pz;
        d1 = (pz != 0.0) ? dd - c * b / pz : d0;
        vx = pz; vy = c;
    }
    double *o = out5 + 5 * (size_t)i;
    o[0] = d.mu[id];
    o[1] = d.mu[id + 1];
    o[2] = atan2(vy, vx);
    o[3] = 2.0 * sqrt(d0 * 5.991);
    o[4] = 2.0 * sqrt(d1 * 5.991);
}

void rekf_launch_ellipses(const RekfDev &d, double *out5, int cap, hipStream_t s)
{
    if (cap <= 0) return;
    hipLaunchKernelGGL(k_ellipses, dim3((cap + 255) / 256), dim3(256), 0, s, d, out5, cap);
}
void rekf_launch_publish_pose(const RekfDev &d, RekfHostSlot *hout, int seq, hipStream_t s)
{
    hipLaunchKernelGGL(k_publish_pose, dim3(1), dim3(64), 0, s, d, hout, seq);
}
