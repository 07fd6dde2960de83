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
// works with (RekfCtl::Rec): ordered compaction by observation, the rank of every matched landmark among the matched ones (ties by
// observation: neighbouring landmarks share cache lines of a column of P, k_mid's gathers run over the row slots in this order), the
// slots of the sub-block, the getters' counts.  `rec` is global memory (the front end's last workgroup, for the k_mid behind the
// kernel boundary) or LDS (every mid workgroup for itself, when the front end runs inside k_mid's own grid).
__device__ static inline void compact_record(RekfCtl::Rec *rec, RekfCtl *ctl, int kind, int oidx, int lane, int K, int n, int n_max, int has_gps)
{
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const unsigned long long ms = __ballot(kind == 1);
    const unsigned long long mm = __ballot(kind == 0);
    const unsigned long long mn = __ballot(kind == 2);
    const int M = __popcll(ms), Mm = __popcll(mm);
    int N2 = __popcll(mn);
    const int room = (n_max - n) / 2;
    if (N2 > room) {                                               // capacity guard (ours)
        if (lane == 0) atomicOr(&ctl->err, REKF_FLAG_CAPACITY);
        N2 = room;
    }
    int rk = 0;
    {
        const int key = (kind == 1) ? oidx : 0x7fffffff;
#pragma unroll
        for (int q = 0; q < 32; ++q) {                               // K <= 32 observations in a whole scan
            const int oq = __builtin_amdgcn_readlane(key, q);
            rk += (oq < key || (oq == key && q < lane)) ? 1 : 0;
        }
    }
    if (kind == 1) {
        const int p = __popcll(ms & lt);
        if (p < 32) {
            rec->pair_obs[p] = lane; rec->pair_id[p] = oidx; rec->pair_state[p] = 1;
            rec->rank[p] = rk;
            rec->urow[2 + rk] = 3 + 2 * oidx; rec->ukc[2 + rk] = 3 + 2 * p;
        }
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
        rec->urow[0] = 0; rec->ukc[0] = 0; rec->urow[1] = 2; rec->ukc[1] = 2;
    }
}
// The front end as a ROLE of a workgroup of NT threads (a multiple of 256): k_front_mb below is nothing else; the fused kernel
// k_dd_front runs it in the workgroups behind its downdate workgroups.  corner_in_ctl: the pose block to predict from is
// RekfCtl::post_C9 (what k_mid evaluated for the previous scan) -- in k_dd_front the previous scan's downdate, which stores that block
// into P, is running beside this role; in k_front_mb it is read from P (set_state, reserve, a flushed host predict may have changed it).
template <int NT>
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
    int n = (d.n_known >= 0) ? d.n_known : ctl->n + (A.aug_pending ? 2 * ctl->n_new : 0);
    if (d.n_known < 0 && A.front_in_mid && A.aug_in_mid) {
        // inside k_mid's grid beside the mid role whose workgroup 0 is appending the previous scan's reflectors (and moving ctl->n) right
        // now: the dimension comes from that scan's augmentation record, as the mid role takes it
        const RekfCtl::AugRec *ar = &ctl->augrec[(A.pred_slot ^ 1) & 1];
        n = ar->n_before + 2 * ar->n2;
    }
    const int L = (n - 3) / 2;
    const int K = A.K;
    // inside k_mid's grid with the motion model evaluated here (one launch per scan): what the mid role reads of this role's Predict goes
    // THROUGH to memory and is complete before this workgroup counts its observation (the mid role's wait for the count orders the rest)
    const bool wt_pred = A.front_in_mid != 0 && !A.host_pred;

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
    if (A.host_pred) {
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
        if (b == 0) for (int q = 0; q < 9; ++q) C9[q] = corner_in_ctl ? ctl->post_C9[q] : rekf_plower(P, (int)ld, q % 3, q / 3);
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
        if (b == 0 && tid == 0) {
            if (A.host_pred) {
#pragma unroll
                for (int q = 0; q < 9; ++q) C9[q] = A.pre_C9[q];
            } else corner_predict(C9, 3, mo);
            RekfCtl::Pred *pr = &ctl->pred[A.pred_slot & 1];
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
        if (!A.host_pred && b == 0 && tid == NT - 64) {        // the wrapped heading (cc:181 / :205), on the last wave
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
        if (mwave == 0 && M_ > 0) {                                // cc:401-425
#pragma clang fp contract(off)
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
            if (kind == 2 && L > 0) {
#pragma clang fp contract(off)
                double g1 = s_part[it & 1][0][0], g2 = s_part[it & 1][0][1]; int gj = s_partj[it & 1][0];
#pragma unroll
                for (int w = 1; w < 4; ++w) argmin2_combine(g1, gj, g2, s_part[it & 1][w][0], s_partj[it & 1][w], s_part[it & 1][w][1]);
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
                if (gj >= 0 && best < 0.6) { kind = 1; best_j = gj; }  // cc:446
            }
            if (mlane == 0) {
                // the result, coherently at agent scope (another workgroup -- on another XCD, behind another L2 -- may be the one that
                // compacts), then the count: whoever sees it reach the scan's target has every result in memory behind it
                __hip_atomic_store(&ctl->obs_kind[i], kind, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ctl->obs_idx[i], best_j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        const int kind = (lane < K) ? __hip_atomic_load(&ctl->obs_kind[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
        const int oidx = (lane < K) ? __hip_atomic_load(&ctl->obs_idx[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
        compact_record(&ctl->rec, ctl, kind, oidx, lane, K, n, d.n_max, A.has_gps);
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
    front_role<1024>(d, A, (int)blockIdx.x, (int)gridDim.x, false);     // (1024 threads: one landmark each to stage; 256 do it in 1.1 us more)
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
template <int NBR>
__device__ static inline bool gj_invert_blocks(v4d (&S)[NBR], int nbr, int w, int g, int c, double (*s_col)[NBR + 1][REKF_PATCH], double *lp)
{
    bool bad = false;
    const v4d zero4 = {0, 0, 0, 0};
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
        lds_barrier();                                // column K and D^-1 of step K are published
        if (w >= nbr) continue;                       // a wave without a block column only keeps the barrier count
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
// ----------------------------------------------------------------------------
// k_mid<NBR>: gather + solve + gain in ONE launch, for at most 16 NBR innovation rows per pass (NBR = 2: up to 16
// matched observations, NBR = 4: up to 32 -- every BASELINE.json configuration; scans with more run several passes,
// see "block step" below).
//
// Round 1 ran k_gather -> k_solve -> k_gain: 24.4 us of kernels at C3 plus two kernel boundaries, of which the
// single-workgroup solve alone was 12.9 us while 255 CUs idled.  The inverse is a LATENCY chain, not work: here
// every workgroup owns 16 state rows and redoes it for itself -- the 67 x 67 sub-block of P that S = H P H^T + Q
// touches is 2.2 k 16-byte loads per workgroup, L2 hits for all but the first -- and then needs neither a launch
// boundary nor a trip through memory for S^-1:
//   A  ordered compaction of the per-observation match results (as k_gather did), H rows packed in LDS;
//   C  all gathers in flight at once: C1 the rows {0,1,2, matched landmark rows} of W = P H^T (for S),
//      C2 this workgroup's 16 rows of W, C3 its 16 columns' (H P)^T, which goes straight to HBM for k_downdate;
//   E  S = H W + Q in the MFMA C layout, blocked Gauss-Jordan (gj_invert_blocks), S^-1 -> LDS;
//   F  K(16 rows) = W S^-1 by MFMA out of LDS; Kn = -K -> HBM; mu += K (z - zhat) -- the reference's own
//      association K_t * (z - z_hat), cc:306 (round 1 formed W (S^-1 dz)) -- pose commit, theta wrap.
// W itself is never written.  All workgroups compute bit-identical S^-1 (same code, same inputs, no atomics).
// Like k_gather it never symmetrises: W from the columns of P, (H P)^T from its rows.
// ----------------------------------------------------------------------------
typedef double v2du __attribute__((ext_vector_type(2), aligned(8)));     // a row pair that starts on an odd row: 8-byte aligned
#define MID_ROWS 16
// MODE picks what else the launch hosts (separate instantiations: the steady state of a full filter, MODE 0, carries none of it --
// the front role's LDS and the extra prologue cost that path 0.6 us per update when they were runtime branches):
//   0  nothing;  1  a filter that can still grow: workgroup 0 leaves the scan's augmentation record (RekfCtl::augrec) and, with
//   A.aug_in_mid, first appends the PREVIOUS scan's new reflectors;  2  the scan's front end runs as the first A.front_in_mid
//   workgroups of this grid (a host-predicted scan behind a pose read-back); also leaves the augmentation record.
//   KCDD > 0 (small states, ONE launch per scan): the PREVIOUS scan's downdate (dd_body<KCDD>, on four of the eight waves) runs as the first
//   A.dd_in_mid workgroups of the grid, the scan's front end (MODE bit 2; the motion model evaluated there) as the next A.front_in_mid; the
//   mid workgroups wait for both inside the launch -- one-way: neither role waits for anybody, their workgroups are dispatched first.
//   MODE there: 2 a full filter, 3 one that can still grow (bit 1: reads / appends the previous scan's augmentation record).
//   dp: the downdate role's device view (that scan's panels); unused -- and never loaded -- when KCDD = 0.
template <int KC> __device__ __forceinline__ void dd_body(const RekfDev &d);
template <int NBR, int MODE, int KCDD = 0>
__global__ __launch_bounds__(512) void k_mid(RekfCtl *ctl_first, RekfDev d, RekfFrontArgs A, RekfDev dp)
{
    constexpr bool FRONT = (MODE & 2) != 0, AUGR = (MODE & 1) != 0, AUGW = MODE >= 1, DDIN = KCDD > 0;
    static_assert(!DDIN || FRONT, "the one-launch form hosts the front end too");
    // ctl_first = d.ctl, as a leading pointer argument of its own: built with -mllvm -amdgpu-kernarg-preload-count the wave starts with it
    // in SGPRs, and the kernel's first loads (the match results) do not wait for the kernel-argument fetch
    constexpr int MP = 16 * NBR;                  // most innovation rows (padded) this instance takes
    constexpr int NPAIR = MP / 2;
    constexpr int NRS = NPAIR + 2;                // row slots of the sub-block: state pairs (sorted by landmark), rows {0,1}, row {2}
    constexpr int NKC = 3 + MP;                   // its columns: 0,1,2, then (col_q, col_q + 1) per pair q
    constexpr int LDS_S = MP + 16;                // row stride of S^-1 in LDS: = 16 mod 32 doubles, so the 4 k-rows of an MFMA operand read hit disjoint banks
    __shared__ double s_col[2][NBR + 1][REKF_PATCH];
    __shared__ double s_leaf[4][REKF_LEAF_SCRATCH];
    __shared__ __attribute__((aligned(16))) double s_coef[8 * MP];          // H row r in 64 bytes, layout of RekfCtl::hrow
    __shared__ __attribute__((aligned(16))) double s_wc0[3][MP];            // rows 0..2 of W
    __shared__ __attribute__((aligned(16))) double s_wcp[NPAIR][MP][2];     // state pair p: its two landmark rows of W, interleaved per column
    __shared__ __attribute__((aligned(16))) double s_wown[MP][MID_ROWS];    // this workgroup's rows of W, [column r][row]; before that, staging of (H P)^T
    // raw P values, as gathered; s_psub is dead once W is formed and S^-1 (phase E) takes its place
    constexpr int NKCP = NKC + 1;                 // sub-block row stride in LDS (even: 16-byte pairs start on odd columns 3 + 2q, read as 8-byte-aligned vectors)
    __shared__ __attribute__((aligned(16))) double s_big[(NKCP * 2 * NRS > MP * LDS_S) ? NKCP * 2 * NRS : MP * LDS_S];
    __shared__ __attribute__((aligned(16))) double s_pw[NKC][MID_ROWS];     // P(own rows, sub-block columns)
    __shared__ double s_dmu[4][MID_ROWS];
    // the scan's match record (RekfCtl::Rec: pairs, ranks, slots, counts, the new reflectors' observation indices = s_cnt[5] of them):
    // a whole scan's comes ready-made from the front end, a block step of a wide scan builds it here
    __shared__ RekfCtl::Rec s_rec;
    int *const s_pair_obs = s_rec.pair_obs, *const s_pair_id = s_rec.pair_id, *const s_pair_state = s_rec.pair_state,
        *const s_rank = s_rec.rank, *const s_cnt = s_rec.cnt, *const s_newid = s_rec.newid;
    __shared__ int s_pcol[NPAIR];
    __shared__ double s_np[3];                    // the committed pose, for their means
    __shared__ double s_dc[4][3][MID_ROWS];       // workgroup 0: partial sums of (K H P)(i, jc), jc = 0..2, per wave of phase F
    // the sub-block's slots in ascending global order -- u = 0: rows / columns {0,1}, u = 1: {2}, u = 2 + rank: a state pair's
    // landmark -- with the first global row (= column) of each and the first sub-block column kc it stands for
    int *const s_urow = s_rec.urow, *const s_ukc = s_rec.ukc;
    static_assert(NRS <= 36 && NPAIR <= 32, "RekfCtl::Rec holds a whole scan");
    double (*s_psub)[NKCP] = (double (*)[NKCP])s_big;                       // [row 2 rs + {0,1} of the sub-block][its column kc]
    double (*s_sinv)[LDS_S] = (double (*)[LDS_S])s_big;                     // S^-1, row-major

#ifdef REKF_DEBUG_TIMING
    long long tqm[16]; int nqm = 0;
    const bool recm = (int)blockIdx.x == 1 + (DDIN ? A.dd_in_mid : 0) + (FRONT ? A.front_in_mid : 0) && threadIdx.x == 0;      // (the second mid workgroup)
    const long long t_entrym = clock64(), w_entrym = wall_clock64();
#define MMARK() do { __builtin_amdgcn_sched_barrier(0); if (recm && nqm < 16) tqm[nqm++] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MMARK()
#endif
    // 512 threads = two teams of four waves (one of each per SIMD).  Waves 0..3 carry the critical chain -- sub-block
    // gather, W rows of S, S, its inverse -- waves 4..7 everything that only this workgroup's 16 rows / columns need
    // (H rows, own gathers, (H P)^T to HBM), off that chain; both meet at the barriers.
    RekfCtl *ctl = ctl_first;
    // A host-predicted scan behind a pose read-back (A.front_in_mid workgroups): its front end -- ReflectorMatch only, pose and pose
    // block came by value -- runs as the FIRST workgroups of this grid instead of a launch of its own (7.5 us + a kernel boundary in
    // front of k_mid, on the path every read-back caller waits for); everybody else waits for the record below.  One-way: the
    // front role waits for nobody, its workgroups are dispatched first and the grid's first 256 workgroups are resident together.
    if constexpr (DDIN) {
        if ((int)blockIdx.x < A.dd_in_mid) {
            // the previous scan's downdate: waves 0..3 (the body is cut for 256 threads; a barrier counts the waves that have not ended).
            // Its tiles go through to memory (DD_STORE); the release covers the plain stores of strips and corner; then the count
            if (threadIdx.x < 256) {
#ifdef REKF_DEBUG_TIMING
                if (threadIdx.x == 0 && blockIdx.x == 0) ctl->dbg[26] = wall_clock64();
#endif
                dd_body<KCDD>(dp);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0) {
#ifdef REKF_DEBUG_TIMING
                    if (blockIdx.x == 0) ctl->dbg[27] = wall_clock64();
#endif
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    (void)__hip_atomic_fetch_add(&ctl->dd_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef REKF_DEBUG_TIMING
                    if (blockIdx.x == 0) ctl->dbg[28] = wall_clock64();
#endif
                }
            }
            return;
        }
    }
    const int bxf = (int)blockIdx.x - (DDIN ? A.dd_in_mid : 0);
    if (FRONT && bxf < A.front_in_mid) {
#ifdef REKF_DEBUG_TIMING
        if (threadIdx.x == 0 && bxf == 0) ctl->dbg[29] = wall_clock64();
#endif
        front_role<512>(d, A, bxf, A.front_in_mid, DDIN);      // (DDIN: the pose block from RekfCtl::post_C9 -- the downdate beside us is storing it into P)
#ifdef REKF_DEBUG_TIMING
        if (threadIdx.x == 0 && bxf == 0) ctl->dbg[30] = wall_clock64();
#endif
        return;
    }
    const int bx = bxf - (FRONT ? A.front_in_mid : 0);          // this workgroup's number among the mid workgroups
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool steam = wave < 4;                  // S team; the other is the "own" team
    const int tt = tid & 255;                     // thread index within the team
    if (FRONT) {
        // every observation's result is in memory once the front end's count has reached the scan's target (each front workgroup
        // writes its result through, drains, then counts): wave 0 polls on one lane, takes the K results past this CU's L1 and
        // compacts them into the record for itself -- no compacting workgroup, no second hand-over
        if (tid < 64) {
            if (tid == 0) {
                unsigned spins = 0;
                while ((int)(__hip_atomic_load(&ctl->front_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - A.front_target) < 0 && ++spins < (1u << 22))
                    __builtin_amdgcn_s_sleep(2);
                if (spins >= (1u << 22)) atomicOr(&ctl->err, REKF_FLAG_STARVED);       // (never alone on the GPU: see rekf_api.hip, in_grid_ok)
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
        if (DDIN && tid == 64) {
            // ... and P is the previous scan's once its downdate role is through (count, then one acquire for the workgroup: what the two
            // roles wrote went through to memory, nothing below needs a coherent load of its own)
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(&ctl->dd_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - A.dd_target) < 0 && ++spins < (1u << 22))
                __builtin_amdgcn_s_sleep(2);
            if (spins >= (1u << 22)) atomicOr(&ctl->err, REKF_FLAG_STARVED);
        }
        __syncthreads();
        MMARK();                                    // (in-grid roles: front end and downdate through)
        if (DDIN) {
            if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __syncthreads();
            MMARK();                                // (... acquired)
        }
    }
    // the scan's match record first, UNCONDITIONALLY (a block step of a wide scan does not use it): a vector load that waits for no
    // scalar one, so the control block costs one memory round trip, not two
    constexpr int NREC = (int)(sizeof(RekfCtl::Rec) / sizeof(int));
    static_assert(NREC <= 512, "one load per thread");
    const int rec_raw = (!FRONT && tid < NREC) ? ((const int *)&ctl->rec)[tid] : 0;
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
        // dispatched first; the wait is bounded all the same and turns into the sticky SINGULAR-free error path: garbage, not a hang)
        if (bx == 0) {
            double *scr = s_big;
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
    const double pose[5] = {hp ? A.pre_pose[0] : ctl->pose_pred[0], hp ? A.pre_pose[1] : ctl->pose_pred[1], hp ? A.pre_pose[2] : ctl->pose_pred[2],
                            hp ? A.pre_pose[3] : ctl->pose_pred[3], hp ? A.pre_pose[4] : ctl->pose_pred[4]};
    const bool pending = FRONT ? true : ctl->pose_pending != 0;      // (the in-grid front role sets it beside us: a host-predicted scan always has one)
    const bool first = bx == 0;
    // the scan's pending Predict (RekfCtl::pred): applied to the gathered P in phase D.  (a, b) = 0 and the pose block as gathered
    // when nothing is pending (later block steps of a wide scan: the first step's downdate has committed it)
    // (through LDS, not registers: eleven uniform doubles held from here to phase D cost this 512-thread kernel its residency)
    const bool do_pred = A.apply_pred != 0;
    __shared__ double s_pred[12];
    if (do_pred && tid >= 64 && tid < 64 + 11) {                                   // ab[0], ab[1], C9[0..8]
        // (with the front role in this grid the control block's copy is being written beside us: a host-predicted scan carries the values)
        const int e = tid - 64;
        s_pred[e] = (FRONT && hp) ? (e < 2 ? A.pre_ab[e] : A.pre_C9[e - 2]) : ((const double *)&ctl->pred[A.pred_slot & 1])[e];
    }

    // ---- A: the scan's matched pairs.  Whole scan (pair0 < 0): ordered compaction of the per-observation results (obs order
    // preserved), wave 0; workgroup 0 also writes the record for the getters.  Block step of a wide scan (pair0 >= 0): the
    // pairs [pair0, pair0 + stride) of the record k_compact_wide wrote, state pairs first, then map pairs.
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
        {
            const int key = st ? id : 0x7fffffff;
#pragma unroll
            for (int q = 0; q < 2 * NPAIR; ++q) {
                const int oq = __builtin_amdgcn_readlane(key, q);
                rk += (oq < key || (oq == key && q < lane)) ? 1 : 0;
            }
        }
        int NSl = M - A.pair0;
        NSl = NSl < 0 ? 0 : (NSl < cnt ? NSl : cnt);
        if (live && lane < NPAIR) {
            s_pair_obs[lane] = ob; s_pair_id[lane] = id; s_pair_state[lane] = st ? 1 : 0;
            if (st) { s_rank[lane] = rk; s_urow[2 + rk] = 3 + 2 * id; s_ukc[2 + rk] = 3 + 2 * lane; }
        }
        if (lane == 0) {
            const bool gps = A.has_gps && cnt > 0 && A.pair0 + cnt == MMtot;     // the pose rows ride on the block step that holds the last pairs
            const int m = (cnt > 0) ? 2 * cnt + (gps ? 3 : 0) : 0;
            s_cnt[0] = cnt; s_cnt[1] = m; s_cnt[2] = (m + 15) & ~15; s_cnt[3] = NSl; s_cnt[4] = gps ? 1 : 0;
            s_cnt[5] = (A.pair0 + A.pair_stride >= A.K) ? ctl->n_new : 0;       // the LAST block step appends the new reflectors' means
            s_urow[0] = 0; s_ukc[0] = 0; s_urow[1] = 2; s_ukc[1] = 2;
        }
    } else if (A.pair0 < 0) {
        // whole scan: the record the front end left (its last workgroup compacted the results, front_role) -- one load, one LDS store
        if (!FRONT && tid < NREC) ((int *)&s_rec)[tid] = rec_raw;                 // (FRONT: compacted above)
    }
#ifdef REKF_DEBUG_MID_FIRST
    MMARK();                                        // (x4: wave 0 through the compaction)
#endif
    __syncthreads();
    MMARK();                                        // 0: compaction done
    const int MM = s_cnt[0], m = s_cnt[1], m_pad = s_cnt[2], NS = s_cnt[3];
    const bool gps_rows = s_cnt[4] != 0;
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
        // ... and what this scan's augmentation needs, should it be deferred into the next scan's k_mid (RekfCtl::augrec)
        if (AUGW && A.K <= REKF_MAX_OBS_DEV) {
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
        }
    };
    // The mean is double-buffered: other workgroups read landmark means from d.mu (phase B) while this one is already
    // done, so the updated rows go to d.mu_out and the host swaps the two pointers behind this launch.
    if (m == 0) {                                   // nothing matched: commit the predicted pose (cc:234 + Predict), no update
        if (tid < MID_ROWS && i0 + tid < n) {
            const int i = i0 + tid;
            const double pp = (i == 0) ? pose[0] : ((i == 1) ? pose[1] : pose[2]);      // no dynamic indexing of pose[]
            const double vv = (pending && i < 3) ? pp : d.mu[i];
            d.mu_out[i] = vv;
            if (first && i < 3) s_np[i] = vv;
        }
        if (pending && first && tid == 0) ctl->pose_pending = 0;
        if (first) {
            // nothing to update: the pose block is the predicted one (or, in a later block step of a wide scan, what the previous
            // step's downdate left in memory)
            if (tid < 9) {
                const int pi = tid % 3, pj = tid / 3, hi = pi > pj ? pi : pj, lo = pi > pj ? pj : pi;
                const double v9 = do_pred ? s_pred[2 + hi + 3 * lo] : rekf_plower(P, (int)ld, hi, lo);
                ctl->post_C9[tid] = v9;
                if (d.pub) host_slot_store(d.pub + 3 + tid, v9, d.pub_seq, 0);
            }
            append_new_means();                 // (a barrier inside: s_np is complete behind it)
            if (d.pub) {                        // (the early publisher, see the end of the kernel)
                if (tid < 3) host_slot_store(d.pub + tid, s_np[tid], d.pub_seq, 0);
                if (tid == 3) host_slot_store(d.pub + 12, (double)(n + 2 * s_cnt[5]), d.pub_seq,
                                              __hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
        }
        // k_downdate2 runs without looking at the control block when the host knows n: give it zeros to add
        const int snb = rekf_strip_base(n);
        for (int e = tid; e < 16 * d.kc_ub; e += 512) {
            const int cidx = e & 15, r = e >> 4, c = i0 + cidx;
            d.HPt[c + (size_t)r * ld] = 0.0;
            d.Kn[c + (size_t)r * ld] = 0.0;
            if (snb >= 0 && c >= snb && c < snb + REKF_STRIP_MAX) { d.HPtB[(c - snb) * REKF_MR_PAD + r] = 0.0; d.KnB[(c - snb) * REKF_MR_PAD + r] = 0.0; }
        }
        return;
    }
    // (m_pad <= MP: the host picked NBR from its bound 2K(+3) of m)

    // ---- C (issued before B: it needs only the match lists): the raw P values.  P is stored as its LOWER triangle (ekf_dev.h).
    // Sub-block P(R, R), R = {0, 1, 2, landmark rows of the state pairs}: only its lower block-triangle is fetched -- blocks
    // (row slot u, column slot u' <= u) in ascending global order, column slot major, so that neighbouring lanes hit neighbouring
    // rows of one column (the 32 nearest reflectors are a handful of runs of consecutive ids): 2 x 16 bytes per block, ~4.7 loads per
    // thread (round 2 fetched the full square: 17) -- and mirrored into the upper block-triangle on the way into LDS.
    // Own rows x R columns: element (i, c) comes from P(i, c) or P(c, i), whichever lies below the diagonal.
    const int nrs = NS + 2, nkc = 3 + 2 * NS;
    const unsigned ldb = (unsigned)d.ld * 8u;                              // bytes per column of P (byte offsets fit 32 bits: ld^2 * 8 < 4 GiB)
    // column of sub-block column kc, for kc = lane and kc = 64 + lane, once: the loops fetch it with v_readlane
    int colA = 0, colB = 0;
    if (lane < nkc) colA = (lane < 3) ? lane : 3 + 2 * s_pair_id[(lane - 3) >> 1] + ((lane - 3) & 1);          // pairs < NS are state pairs
    if (64 + lane < nkc) colB = 3 + 2 * s_pair_id[(61 + lane) >> 1] + ((61 + lane) & 1);
    constexpr int NBLK = NRS * (NRS + 1) / 2, PS_IT = (NBLK + 255) / 256;
    v2du ps[PS_IT][2];
    int blk_u[PS_IT], blk_v[PS_IT];                                         // row slot u, column slot u' of this thread's blocks (-1: none)
    const int nblk = nrs * (nrs + 1) / 2;
    if (steam) {
#pragma unroll
        for (int it = 0; it < PS_IT; ++it) {
            const int t = tt + 256 * it;
            const int tc = (t < nblk) ? t : nblk - 1;                       // clamped: a thread past the end refetches the last block
            // column slot v of block tc in the column-major enumeration of the lower block-triangle: v columns hold
            // v nrs - v (v - 1) / 2 blocks
            // (the hardware's approximate square root + 24-bit multiplies here: measured 0.65 us SLOWER per update, A/B in one session)
            const float bq = 2.0f * (float)nrs + 1.0f;
            int v = (int)((bq - sqrtf(fmaxf(bq * bq - 8.0f * (float)tc, 0.0f))) * 0.5f);
            v = max(0, min(nrs - 1, v));
            auto c0 = [&](int vv) __attribute__((always_inline)) { return vv * nrs - (vv * (vv - 1)) / 2; };   // blocks in columns < vv
            while (v > 0 && c0(v) > tc) --v;
            while (v + 1 < nrs && c0(v + 1) <= tc) ++v;
            const int u = v + (tc - c0(v));
            blk_u[it] = (t < nblk) ? u : -1; blk_v[it] = v;
            const char *rp = (const char *)(P + s_urow[u]) + (unsigned)s_urow[v] * ldb;
            ps[it][0] = *(const v2du *)rp;                                  // rows (r, r+1) of column c
            ps[it][1] = *(const v2du *)(rp + ldb);                          // ... of column c + 1
        }
    }
    constexpr int PW_IT = (NKC * 8 + 255) / 256;
    v2d pw[PW_IT];
    if (!steam) {
        const int pr = tid & 7, sub = (tid >> 3) & 7;                       // 8 columns x 8 row pairs per wave instruction
        const int kc0 = __builtin_amdgcn_readfirstlane(wave & 3);
        const int i = i0 + 2 * pr;
#pragma unroll
        for (int it = 0; it < PW_IT; ++it) {
            int kc = 8 * (kc0 + 4 * it) + sub;
            kc = (kc < nkc) ? kc : nkc - 1;
            const int cA = __shfl(colA, kc & 63, 64), cB = __shfl(colB, kc & 63, 64);
            const int cc = (kc < 64) ? cA : cB;
            // (i, cc) and (i + 1, cc), each from below the diagonal
            const double *e0 = (i >= cc) ? P + (size_t)i + (size_t)cc * ld : P + (size_t)cc + (size_t)i * ld;
            const double *e1 = (i + 1 >= cc) ? P + (size_t)(i + 1) + (size_t)cc * ld : P + (size_t)cc + (size_t)(i + 1) * ld;
            pw[it].x = *e0; pw[it].y = *e1;
        }
    }
    MMARK();                                        // 1: gathers issued

    // ---- B: H rows (cc:248-304, gps.cc:305-332), thread = row (own team: its landmark-mean loads overlap the S team's gather)
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
            double corr = hr[0] * (d.mu[0] - pose[0]);
            corr += hr[1] * (d.mu[1] - pose[1]);
            corr += hr[2] * dth;
            if (col >= 0) {
                corr += hr[3] * (d.mu[col] - d.mu_lin[col]);
                corr += hr[4] * (d.mu[col + 1] - d.mu_lin[col + 1]);
            }
            hr[6] = hr[6] - corr;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) s_coef[8 * r + q] = hr[q];
        if (rr == 0) s_pcol[p] = col;
    }
    // raw values -> LDS.  s_psub is [row 2 rs + {0,1}][sub-block column kc] (rs = the pair's rank among the state pairs; NS for
    // rows {0,1}, NS + 1 for row {2}): phase D reads a row pair's (col_q, col_q + 1) as ONE 16-byte value, consecutive q =
    // consecutive addresses (no bank conflicts).  A block goes in twice: as fetched and transposed (the upper block-triangle).
    if (steam) {
#pragma unroll
        for (int it = 0; it < PS_IT; ++it) {
            const int u = blk_u[it], v = blk_v[it];
            if (u >= 0) {
                const int ru = (u >= 2) ? u - 2 : NS + u, rv = (v >= 2) ? v - 2 : NS + v;       // their row slots in s_psub
                const int ku = s_ukc[u], kv = s_ukc[v];
                double b00 = ps[it][0].x, b10 = ps[it][0].y, b01 = ps[it][1].x, b11 = ps[it][1].y;   // b[a][e] = P(r + a, c + e)
                if (s_urow[u] == s_urow[v]) b01 = b10;                      // a diagonal block (also: two pairs on ONE landmark, Q6): (r, r+1) lies above the diagonal
                // slot 1 stands for row / column 2 ALONE: its second row (3) is fetched but never used, its second column does not exist
                s_psub[2 * ru][kv] = b00; s_psub[2 * ru + 1][kv] = b10;
                if (v != 1) { s_psub[2 * ru][kv + 1] = b01; s_psub[2 * ru + 1][kv + 1] = b11; }
                if (u != v) {
                    s_psub[2 * rv][ku] = b00;
                    if (v != 1) s_psub[2 * rv + 1][ku] = b01;
                    if (u != 1) { s_psub[2 * rv][ku + 1] = b10; if (v != 1) s_psub[2 * rv + 1][ku + 1] = b11; }
                }
            }
        }
    } else {
        const int pr = tid & 7, sub = (tid >> 3) & 7;
        const int kc0 = __builtin_amdgcn_readfirstlane(wave & 3);
#pragma unroll
        for (int it = 0; it < PW_IT; ++it) {
            const int kc = 8 * (kc0 + 4 * it) + sub;
            if (kc < nkc) *(v2d *)&s_pw[kc][2 * pr] = pw[it];
        }
    }
    __syncthreads();
    MMARK();                                        // 2: H rows and raw P in LDS

    // (phase E multiplies the landmark rows of W of a CLAMPED pair by the zero coefficients of rows that have no landmark block:
    // with no state pair at all -- a scan that matched the pre-loaded map only -- that pair does not exist, and whatever the
    // last kernel left in LDS there must not be a NaN)
    if (NS == 0 && tid < MP) { s_wcp[0][tid][0] = 0.0; s_wcp[0][tid][1] = 0.0; }
    // ---- D: form W (rows of S, own rows) and (H P)^T (own columns) out of LDS.
    // Same operation order as k_gather: v = p0 h0; v += p1 h1; v += p2 h2; v += pl0 g0; v += pl1 g1
    const int nq = m_pad / 2;                        // row pairs, pad rows included (their H rows are zero)
    const int strip_nb = rekf_strip_base(n);          // k_downdate's border strips want rows nb.. of HPt / Kn contiguous
    if (steam) {
        // items (slot, q): slot < NS = state pair (its two landmark rows), slot NS = rows 0,1, slot NS+1 = row 2 (its second
        // value is row 3: computed, never stored).  q = tid mod NPAIR, slot = tid / NPAIR + (256 / NPAIR) pass
        const int q = tt & (NPAIR - 1);
        const bool qlive = q < nq;
        const bool has_col = qlive && s_pcol[q] >= 0; // a state pair: q < NS, its columns are kc = 3 + 2q, 4 + 2q
        const int kq = has_col ? 3 + 2 * q : 0;       // clamped: without a landmark block the two extra terms are multiplied by zeros
        double h0[5], h1[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) { h0[k] = s_coef[16 * q + k]; h1[k] = s_coef[16 * q + 8 + k]; }
        if (!has_col) { h0[3] = 0; h0[4] = 0; h1[3] = 0; h1[4] = 0; }
        constexpr int SL_STEP = 256 / NPAIR, SL_PASS = (NRS + SL_STEP - 1) / SL_STEP;
        int rs2[SL_PASS];
#pragma unroll
        for (int pass = 0; pass < SL_PASS; ++pass) {
            const int slot = tt / NPAIR + SL_STEP * pass, sl = (slot < nrs) ? slot : nrs - 1;
            rs2[pass] = 2 * ((sl < NS) ? s_rank[sl] : sl);
        }
#pragma unroll
        for (int pass = 0; pass < SL_PASS; ++pass) {
            const int slot = tt / NPAIR + SL_STEP * pass;
            v2d a01 = *(const v2d *)&s_psub[rs2[pass]][0], b01 = *(const v2d *)&s_psub[rs2[pass] + 1][0];
            double a2 = s_psub[rs2[pass]][2], b2 = s_psub[rs2[pass] + 1][2];
            v2d al = *(const v2du *)&s_psub[rs2[pass]][kq], bl = *(const v2du *)&s_psub[rs2[pass] + 1][kq];
            if (do_pred) {
                // the pending Predict, G P G^T + V restricted to the sub-block (G = I + a e0 e2^T + b e1 e2^T): a landmark row takes
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
                    const v2d r2 = *(const v2du *)&s_psub[2 * (NS + 1)][kq];           // P(2, c), P(2, c + 1)
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
    } else {
        // own rows: item (row pair pr, q), q = tt / 8 (+ 32 per pass)
        const int pr = tt & 7;
#pragma unroll
        for (int pass = 0; pass < (NPAIR + 31) / 32; ++pass) {
            const int q2 = (tt >> 3) + 32 * pass;
            if (q2 < nq) {
                const bool hc = s_pcol[q2] >= 0;
                const int c = i0 + 2 * pr;
                v2d p0 = *(const v2d *)&s_pw[0][2 * pr], p1 = *(const v2d *)&s_pw[1][2 * pr], p2 = *(const v2d *)&s_pw[2][2 * pr];
                v2d l0 = {0, 0}, l1 = {0, 0};
                if (hc) { l0 = *(const v2d *)&s_pw[3 + 2 * q2][2 * pr]; l1 = *(const v2d *)&s_pw[4 + 2 * q2][2 * pr]; }
                if (do_pred) {                        // the pending Predict on this workgroup's rows (see the S team above)
#pragma clang fp contract(off)
                    const double pa = s_pred[0], pb = s_pred[1];
                    const double *pC9 = s_pred + 2;
                    if (c >= 3) {
                        p0.x = p0.x + pa * p2.x; p1.x = p1.x + pb * p2.x;
                        p0.y = p0.y + pa * p2.y; p1.y = p1.y + pb * p2.y;
                    } else if (c == 0) {              // rows 0, 1 (workgroup 0): the predicted block; landmark columns against row 2's
                        p0.x = pC9[0]; p0.y = pC9[1]; p1.x = pC9[3]; p1.y = pC9[4]; p2.x = pC9[6]; p2.y = pC9[7];
                        if (hc) {
                            const double r20 = s_pw[3 + 2 * q2][2], r21 = s_pw[4 + 2 * q2][2];
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
                    if (strip_nb >= 0) {
                        if (c >= strip_nb && c < strip_nb + REKF_STRIP_MAX) d.HPtB[(c - strip_nb) * REKF_MR_PAD + r] = vx;
                        if (c + 1 >= strip_nb && c + 1 < strip_nb + REKF_STRIP_MAX) d.HPtB[(c + 1 - strip_nb) * REKF_MR_PAD + r] = vy;
                    }
                }
            }
        }
        // columns [m_pad, kc_ub) of HPt / Kn are kept zero for k_downdate2<kc_ub>
        for (int e = tt; e < 16 * (d.kc_ub - m_pad); e += 256) {
            const int cidx2 = e & 15, r = m_pad + (e >> 4), c2 = i0 + cidx2;
            d.HPt[c2 + (size_t)r * ld] = 0.0;
            d.Kn[c2 + (size_t)r * ld] = 0.0;
            if (strip_nb >= 0 && c2 >= strip_nb && c2 < strip_nb + REKF_STRIP_MAX) {
                d.HPtB[(c2 - strip_nb) * REKF_MR_PAD + r] = 0.0;
                d.KnB[(c2 - strip_nb) * REKF_MR_PAD + r] = 0.0;
            }
        }
    }
    __syncthreads();
    MMARK();                                        // 3: W rows in LDS, (H P)^T stored

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
        const bool bad = gj_invert_blocks<NBR>(S, nbr, w, g, c, s_col, s_leaf[w & 3]);
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
                    if (strip_nb >= 0 && i0 + idx >= strip_nb && i0 + idx < strip_nb + REKF_STRIP_MAX)
                        d.KnB[(i0 + idx - strip_nb) * REKF_MR_PAD + j] = -acc[r];
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
        if (i < n) {
            double dm = 0.0;
#pragma unroll
            for (int jt = 0; jt < NBR; ++jt) dm += s_dmu[jt][tid];
            const double pp = (i == 0) ? pose[0] : ((i == 1) ? pose[1] : pose[2]);
            const double base = (pending && i < 3) ? pp : d.mu[i];
            double v = base + dm;
            if (i == 2) v = atan2(sin(v), cos(v));                               // cc:307
            d.mu_out[i] = v;
            if (i == 0) ctl->pose_pending = 0;
            if (first && i < 3) s_np[i] = v;
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
            ctl->post_C9[e] = v9;
            if (d.pub) host_slot_store(d.pub + 3 + e, v9, d.pub_seq, 0);
        }
        append_new_means();                     // (a barrier inside: s_np is complete behind it)
        // The EARLY publisher (d.pub set on this launch: the caller has been reading the pose back after its scans, rekf_api.hip): pose,
        // block, n and flags go to the host from here, a kernel before the downdate -- at the price of 0.8 us at the end of this kernel
        // (it ends when the PCIe writes are through).  Otherwise the downdate's first workgroup publishes, at its start.
        if (d.pub) {
            if (tid < 3) host_slot_store(d.pub + tid, s_np[tid], d.pub_seq, 0);
            if (tid == 3) host_slot_store(d.pub + 12, (double)(n + 2 * s_cnt[5]), d.pub_seq,
                                          __hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
    }
#if defined(REKF_DEBUG_TIMING) && !defined(REKF_DEBUG_DD2)
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
//
// Border strips.  When n is a few rows past a multiple of the tile size (n = 3 + 2L with L a multiple of 32: three
// rows), a last tile row/column would be 95 % padding yet cost a full tile of traffic and MFMA time -- and with
// 33^2 = 1089 tiles on 256 workgroups, a FIFTH tile for a quarter of them.  Instead the tile grid covers [0, nb)^2,
// nb = 64 floor(n / 64), and the strips P(nb.., :) and P(:, nb..) ride on the diagonal tiles: the workgroup that has the
// panels Kn(I,:) and HPt(I,:) of tile (I,I) in LDS also updates P(64I.., nb..) and P(nb.., 64I..) with plain FMAs
// against the border rows of Kn / HPt (KnB / HPtB, staged in s_border).
// ----------------------------------------------------------------------------
#define DT 64
#define DD_WG_PER_CU 1
#define DD_STRIP_MAX REKF_STRIP_MAX
typedef double DdBorder[DD_STRIP_MAX][REKF_MR_PAD];

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
// The body is shared by two kernels: k_downdate2 (the downdate alone) and k_dd_front (LAZY DOWNDATE, rekf_api.hip: the downdate of scan
// t enqueued with scan t+1, whose front end -- Predict's pose and ReflectorMatch, which need the mean and the pose block k_mid(t) left
// but nothing else of P -- runs in extra workgroups beside it).
template <int KC>
__device__ __forceinline__ void dd_body(const RekfDev &d)
{
    extern __shared__ __attribute__((aligned(16))) double dd_smem[];   // [Kn 0 | Kn 1 | HPt 0 | HPt 1] panels (+ 16 KiB strip scratch if KC < 64)
    __shared__ __attribute__((aligned(16))) double s_border[2][DD_STRIP_MAX][REKF_MR_PAD];
#ifdef REKF_DEBUG_TIMING
    __shared__ long long tq2[24];          // (in LDS: a register array costs every thread 48 VGPRs and changes what is being measured)
    int nq2 = 0;
#ifdef REKF_DEBUG_DD2
#ifndef REKF_DEBUG_DD2_BLOCK
#define REKF_DEBUG_DD2_BLOCK 0
#endif
    const bool rec2 = blockIdx.x == REKF_DEBUG_DD2_BLOCK && threadIdx.x == 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) const_cast<RekfCtl *>(d.ctl)->dbg[3] = wall_clock64();   // block 0's entry, for the offset of the recorded block
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
    // (through LDS like the border rows: requested now, written behind the prologue's DMA wait)
    const bool pred_on = d.pred_slot >= 0;
    __shared__ double s_pred[12];
    double pred_v = 0.0;
    if (pred_on && threadIdx.x >= 64 && threadIdx.x < 64 + 11) pred_v = ((const double *)&ctl->pred[d.pred_slot & 1])[threadIdx.x - 64];   // ab[0], ab[1], C9[0..8]
    // the pose block after this update, as k_mid evaluated and published it (RekfCtl::post_C9): tile (0, 0) stores THOSE bits
    __shared__ double s_post[9];
    double post_v = 0.0;
    if (blockIdx.x == 0 && threadIdx.x >= 128 && threadIdx.x < 128 + 9) post_v = ctl->post_C9[threadIdx.x - 128];
    int n = d.n_known;
    if (n < 0) {
        n = ctl->n;
        // (with a Predict pending the kernel runs even so: k_mid has left zero panels, and the tiles of column 0 commit the Predict)
    }
    // The publisher (the scan's last downdate): pose mean, the pose block AFTER this update (RekfCtl::post_C9: k_mid evaluated it; tile
    // (0, 0) below stores the same bits), the n the state will have once the k_augment behind this kernel has run, and the flags --
    // at the START of the kernel: everything is known, and the stores are long through when the kernel ends
    if (d.pub && blockIdx.x == 0) {
        const int l = threadIdx.x;
        if (l < 3) host_slot_store(d.pub + l, d.mu[l], d.pub_seq, 0);
        else if (l < 12) host_slot_store(d.pub + l, ctl->post_C9[l - 3], d.pub_seq, 0);
        else if (l == 12) host_slot_store(d.pub + 12, (double)(n + (d.pub_aug ? 2 * ctl->n_new : 0)), d.pub_seq,
                                          __hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    if (d.n_known < 0 && ctl->m == 0 && !pred_on) return;   // nothing matched and nothing pending: P stays as it is
    constexpr int NK = KC / 4;               // MFMA k-steps per tile
    constexpr int ND = KC / 8;               // DMA instructions per panel per wave (2 k-rows each, 4 waves)
    constexpr int PANEL = KC * 64;           // doubles per panel
    static_assert(KC % 16 == 0 && KC >= 16 && KC <= 64, "one k-chunk");
    constexpr int Q4 = NK / 4;               // k-steps per VMEM phase: [DMA Kn][stores (+ DMA HPt)][P loads][-]
    const int rem = n % DT;
    const bool strips = rem > 0 && rem <= DD_STRIP_MAX && n >= DT;
    const int T = strips ? n / DT : (n + DT - 1) / DT;
    int w = blockIdx.x;
    const int nw = d.dd_grid > 0 ? d.dd_grid : (int)gridDim.x;      // (k_dd_front: the workgroups behind dd_grid are the front end)
    // the triangle column by column (tile column J: I = J .. T-1); an XCD's workgroups take consecutive ranges of it
    if (nw >= 8 && (nw & 7) == 0) w = (blockIdx.x & 7) * (nw >> 3) + (blockIdx.x >> 3);
    // Two classes of workgroups.  Class A, workgroups [0, T): the diagonal tile (w, w) -- it costs almost two
    // ordinary tiles: no transposed image, but the border strips and the symmetric finish -- behind the tile below it, (w+1, w),
    // which shares its HPt panel.  Class B, the rest: the tiles with I >= J + 2, column by column, in equal ranges (3 per
    // workgroup at T = 32).  With the diagonal tiles inside equal ranges of three, the workgroups that held one set the pace.
    // (When class B gets fewer than three tiles per workgroup -- small states -- class A keeps to its diagonal tile and the tiles
    // below the diagonal join class B: dd_sub = 1.)
    const bool classA = w < T;

    // class B: every free workgroup takes tiles -- lo each, the first x of them one more (2 or 3 at T = 32: 465 tiles on 224
    // workgroups; round 2 gave three tiles to 155 workgroups and left 69 CUs idle).  (lo, x) come from the host when it knows n
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
    if (t_begin >= t_end) return;
    const int nt = t_end - t_begin;
    const size_t ld = (size_t)d.ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int idx = lane & 15, kq = lane >> 4;
    const int wi = wave & 1, wj = wave >> 1;
    const double *__restrict__ Kn = d.Kn;
    const double *__restrict__ HPt = d.HPt;
    double *__restrict__ P = d.P;

    // (a class-A workgroup takes its diagonal tile LAST: the strip work then rides on a tile that has nothing to prefetch)
    const int TT = T - sub;                                 // side of the triangle class B enumerates (I >= J + sub)
    // tile number -> (row, column) of the triangle, with a cursor (column, its first tile number) that moves to the queried tile:
    // a workgroup asks for a handful of neighbouring tiles, so after the first query (<= T scalar steps) every look-up is O(1).
    // (A closed form with sqrtf + fix-up loops, evaluated afresh for each of the ~8 look-ups of the prologue, cost 0.8 us.)
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
    auto p_ptr = [&](int I, int J) __attribute__((always_inline)) -> double * {
        return P + (size_t)(DT * I + 32 * wi + 2 * idx) + (size_t)(DT * J + 32 * wj + 2 * kq) * ld;
    };
    // P block of a tile -> P + acc -> store source.  NB register blocks in rotation: the read stream runs AHEAD tiles ahead
    // of the MFMAs.  Measured at C3 (4 tiles per workgroup): AHEAD = 1 and 2 give the same kernel time (17.7 us) -- with two
    // blocks in flight the first MFMA loop stretches from 2.3 to 3.6 us: a wave that cannot issue its load (memory queue
    // full) cannot issue its MFMAs either.
#ifndef REKF_DD_AHEAD
#define REKF_DD_AHEAD 1
#endif
    constexpr int AHEAD = REKF_DD_AHEAD, NB = AHEAD + 1;
    v2d pq[NB][8];
    v4d acc[2][2];
    int I, J, kb = 0, hb = 0;
    tile_IJ(0, I, J);
    // The tile list of a short range, ONCE, in scalar registers: the straight-line forms below index it with compile-time
    // positions.  (Until round 3 every tile body looked its neighbours up through the cursor -- three look-ups per tile, each a
    // pair of scalar loops -- and the serial scalar code between two MFMA loops cost a lone wave per SIMD most of a microsecond.)
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

    // ---- border strips (see "Border strips" above): same scheme as before, on whichever tile is diagonal
    auto strip_addr = [&](int II, double *&p0, double *&p1, int &which, int &b, int &x) __attribute__((always_inline)) {
        which = tid / (DD_STRIP_MAX * 32); const int u = tid % (DD_STRIP_MAX * 32);
        b = u >> 5; x = 2 * (u & 31);
        const int nb = DT * T;
        if (which == 0) { p0 = P + (size_t)(DT * II + x) + (size_t)(nb + b) * ld; p1 = p0 + 1; }
        else { p0 = P + (size_t)(nb + b) + (size_t)(DT * II + x) * ld; p1 = p0 + ld; }
    };
    static_assert(2 * DD_STRIP_MAX * 32 <= 256, "one pass of the strip mapping");

    D2MARK();                                // 0: ctl read, tile assignment done
    // ---- prologue: border rows, both panels of tile 0 by DMA, its P block.  Only a class-A workgroup (the one with a diagonal
    // tile) needs the border rows, and their way into LDS must not stand in front of the DMA: the loads go out first, the
    // ds_writes wait behind the DMA issue (round 2 had load -> wait -> ds_write -> DMA: a whole memory round trip, for every
    // workgroup, before its first panel was even requested)
    v2d bdr0 = {0, 0}, bdr1 = {0, 0};
    const bool want_border = strips && classA;
    if (want_border) { bdr0 = ((const v2d *)d.KnB)[tid]; bdr1 = ((const v2d *)d.HPtB)[tid]; }
#pragma unroll
    for (int q = 0; q < ND; ++q) dma_piece(Kn, DT * I, 0, q);
#pragma unroll
    for (int q = 0; q < ND; ++q) dma_piece(HPt, DT * J, 2, q);
    {
        const double *Pw = p_ptr(I, J);
#pragma unroll
        for (int q = 0; q < 8; ++q) pq[0][q] = DD_LOAD((const v2d *)(Pw + (size_t)(8 * (q & 3) + (q >> 2)) * ld));
    }
    if (AHEAD > 1 && nt > 1) {               // ... and the P block of tile 1
        int I1, J1;
        tile_IJ(1, I1, J1);
        const double *Pw = p_ptr(I1, J1);
#pragma unroll
        for (int q = 0; q < 8; ++q) pq[NB - 2][q] = DD_LOAD((const v2d *)(Pw + (size_t)(8 * (q & 3) + (q >> 2)) * ld));
        dd_wait_vmcnt<16>();                 // the DMAs (and everything before them); the 16 P loads may still fly
    } else dd_wait_vmcnt<8>();
    if (pred_on && tid >= 64 && tid < 64 + 11) s_pred[tid - 64] = pred_v;
    if (blockIdx.x == 0 && tid >= 128 && tid < 128 + 9) s_post[tid - 128] = post_v;
    if (want_border) {                       // (older than the DMAs: arrived)
        ((v2d *)&s_border[0][0][0])[tid] = bdr0;                   // s_border[0] = Kn rows nb.., [1] = HPt rows nb..
        ((v2d *)&s_border[1][0][0])[tid] = bdr1;
    }
    lds_barrier();
    D2MARK();                                // 1: panels of tile 0 landed

    // ---- one tile.  PAR = which third of pq holds this tile's P block; FIRST (no tile before it: nothing to store), LOAD2
    // (tile pos+2 exists: request its P block), LAST and SPECIAL are compile-time so that every load and store of a variant
    // is unconditional: hipcc's s_waitcnt pass then counts them exactly (with `if (pos > 0)` around the stores it had to
    // assume the fewest, and waited vmcnt(0) for a P block it had only just requested).  SPECIAL = diagonal tile that carries
    // the border strips; it prefetches no panels (its idle Kn buffer is the strip scratch), which costs nothing when it is
    // the last tile -- the usual case.
    auto tile_body = [&](auto par_c, auto first_c, auto load2_c, auto last_c, auto special_c, auto pos) __attribute__((always_inline)) {
        using Plus1 = std::integral_constant<int, 1>;
        using PlusA = std::integral_constant<int, AHEAD>;
        using Minus1 = std::integral_constant<int, -1>;
        constexpr int PAR = decltype(par_c)::value, PREV = (PAR + AHEAD) % NB;   // PREV: tile pos-1's block = where tile pos+AHEAD's goes
        constexpr bool FIRST = decltype(first_c)::value, LOAD2 = decltype(load2_c)::value, LAST = decltype(last_c)::value,
                       SPECIAL = decltype(special_c)::value == 2,          // diagonal tile that carries the border strips
                       DIAGSYM = decltype(special_c)::value >= 1;          // diagonal tile: its upper half := mirror of its (new) lower half
        static_assert(!(LAST && LOAD2), "no tile after the last");
        static_assert(!DIAGSYM || LAST, "a diagonal tile ends its workgroup's range (class A)");
        int In = I, Jn = J;
        if constexpr (!LAST) coords(shift(pos, Plus1()), In, Jn);
        const bool needK = !LAST && In != I, needH = !LAST && Jn != J;
        const double *Pn = nullptr;                         // tile pos+2's P block
        if constexpr (LOAD2) { int I2, J2; coords(shift(pos, PlusA()), I2, J2); Pn = p_ptr(I2, J2); }
        double *Po = nullptr;                               // where tile pos-1 goes
        if constexpr (!FIRST) { int Ip, Jp; coords(shift(pos, Minus1()), Ip, Jp); Po = p_ptr(Ip, Jp); }
        const double *aW = hp_buf(hb) + 32 * wj + 2 * idx + kq * 64;        // A[j][k] = HP(k,j)
        const double *bK = kn_buf(kb) + 32 * wi + 2 * idx + kq * 64;        // B[k][i] = Kn(i,k)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0, 0, 0, 0};
        // strip operands of a special tile: lane -> (strip, x), wave -> quarter of the k range
        const int s_which = lane >> 5, s_x = 2 * (lane & 31), s_kb = (KC / 4) * wave;
        const double *s_panel = (s_which ? hp_buf(hb) : kn_buf(kb)) + s_x + s_kb * 64;
        const double *s_brow = &s_border[s_which ? 0 : 1][0][s_kb];
        v2d sacc[DD_STRIP_MAX];
        v2d strip_p = {0.0, 0.0};
        if (SPECIAL) {
            double *p0, *p1; int which, b, x;
            strip_addr(I, p0, p1, which, b, x);
            if (which == 1 && b < rem) { strip_p.x = *p0; strip_p.y = *p1; }      // the ROW strip P(nb.., 64 I ..): the column strip lies above the diagonal
#pragma unroll
            for (int b2 = 0; b2 < DD_STRIP_MAX; ++b2) { sacc[b2].x = 0.0; sacc[b2].y = 0.0; }
        }
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
            if (SPECIAL && (kk & 1) == 0) {                 // strip FMAs ride under the MFMAs: k pair (kk, kk+1) of this wave's quarter
                const v2d v0 = *(const v2d *)(s_panel + kk * 64), v1 = *(const v2d *)(s_panel + (kk + 1) * 64);
#pragma unroll
                for (int b3 = 0; b3 < DD_STRIP_MAX; ++b3) {
                    const v2d bb = *(const v2d *)(s_brow + b3 * REKF_MR_PAD + kk);
                    sacc[b3].x = fma(v0.x, bb.x, sacc[b3].x); sacc[b3].y = fma(v0.y, bb.x, sacc[b3].y);
                    sacc[b3].x = fma(v1.x, bb.y, sacc[b3].x); sacc[b3].y = fma(v1.y, bb.y, sacc[b3].y);
                }
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
            // its range): S takes the front of the dynamic LDS (33 KiB; the smallest launch has 48 KiB).
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
            lds_barrier();                                  // (the strip reduction below reuses LDS)
        }
        if (SPECIAL) {
            // partial strip sums of the four waves meet in the idle Kn buffer; the threads of the strip mapping finish
            // (with KC < 64 a panel is smaller than the 16 KiB of partial sums: the launch's LDS beyond the four panels is free)
            v2d *red = (v2d *)(dd_smem + (KC == 64 ? (size_t)(kb ^ 1) * PANEL : (size_t)4 * PANEL));
#pragma unroll
            for (int b3 = 0; b3 < DD_STRIP_MAX; ++b3) red[(wave * DD_STRIP_MAX + b3) * 64 + lane] = sacc[b3];
            lds_barrier();
            {
                double *p0, *p1; int which, b, x;
                strip_addr(I, p0, p1, which, b, x);
                // (a pending Predict on the row strip of tile column 0: columns 0, 1 in the thread x = 0, column 2 in its neighbour)
                const double s_c2 = __shfl_down(strip_p.x, 1, 64);
                if (pred_on && I == 0 && which == 1 && b < rem && x == 0) {
#pragma clang fp contract(off)
                    strip_p.x = strip_p.x + s_pred[0] * s_c2; strip_p.y = strip_p.y + s_pred[1] * s_c2;
                }
                if (which == 1 && b < rem) {
                    v2d t = strip_p;
#pragma unroll
                    for (int w4 = 0; w4 < 4; ++w4) { const v2d r = red[(w4 * DD_STRIP_MAX + b) * 64 + 32 + (x >> 1)]; t.x += r.x; t.y += r.y; }
                    *p0 = t.x; *p1 = t.y;
                }
            }
            if (I == 0 && tid < 64) {                   // the corner block P(nb.., nb..): 16 (a,b) slots x 4 quarters of k
                const int a = (tid >> 2) & 3, b = tid & 3, k4 = tid >> 4;
                const int as = max(a, b), bs = min(a, b);      // (a,b) and (b,a) both take the lower element's sum
                const double *ra = &s_border[0][as][(KC / 4) * k4], *rb = &s_border[1][bs][(KC / 4) * k4];
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < KC / 4; k += 2) {
                    const v2d u = *(const v2d *)(ra + k), ww = *(const v2d *)(rb + k);
                    v = fma(u.x, ww.x, v); v = fma(u.y, ww.y, v);
                }
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (k4 == 0 && a < rem && b < rem && a >= b) {      // lower triangle of the corner block
                    double *pp = P + (size_t)(DT * T + a) + (size_t)(DT * T + b) * ld;
                    *pp += v;
                }
            }
        }
        D2MARK();                            // P block there, combined (+ strips)
        if (LAST) {
            double *Pw = p_ptr(I, J);
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
    using C0 = std::integral_constant<int, 0>;      // tile kinds: off-diagonal / diagonal / diagonal with the border strips
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    // Short ranges (the BASELINE sizes: 4 tiles per workgroup at n = 2051, 1 at n = 1027 / 259) run as STRAIGHT-LINE code,
    // one instantiation per position: with no loop and no join in the way, hipcc's s_waitcnt pass places every wait exactly
    // (through the generic loop below it merges the variants' states at the joins and waits for far younger loads than the
    // P block it needs).  A diagonal tile ends its (class A) range, so only the last position has the diagonal variants.
    // The two classes take separate code: class A ends on its diagonal tile (mirror, strips, publication), class B never sees one.
    auto straight = [&](auto nt_c, auto cls_a) __attribute__((always_inline)) {
        constexpr int NT = decltype(nt_c)::value;
        constexpr bool CLS_A = decltype(cls_a)::value;
        auto one = [&](auto pos_c) __attribute__((always_inline)) {
            constexpr int POS = decltype(pos_c)::value;
            using Par = std::integral_constant<int, POS % NB>;
            using First = std::integral_constant<bool, POS == 0>;
            using Load2 = std::integral_constant<bool, (POS + AHEAD < NT)>;
            using Last = std::integral_constant<bool, POS == NT - 1>;
            if constexpr (POS == NT - 1 && CLS_A) {
                if (strips) tile_body(Par(), First(), Load2(), Last(), C2(), pos_c);
                else tile_body(Par(), First(), Load2(), Last(), C1(), pos_c);
            } else tile_body(Par(), First(), Load2(), Last(), C0(), pos_c);
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
    dd_body<KC>(d);
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
    dd_body<KC>(d);
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
void rekf_launch_mid(const RekfDev &d, const RekfFrontArgs &a, int n_ub, int m_ub, bool mode_grow, hipStream_t s)
{
    // m_ub <= 64 (the host checks): one workgroup per 16 state rows
    const int grid = (n_ub + MID_ROWS - 1) / MID_ROWS + (a.front_in_mid > 0 ? a.front_in_mid : 0);
    const int mode = (a.front_in_mid > 0) ? 2 : (mode_grow ? 1 : 0);
    if (m_ub <= 32) {
        if (mode == 0) hipLaunchKernelGGL((k_mid<2, 0>), dim3(grid), dim3(512), 0, s, d.ctl, d, a, d);          // (the last argument: the downdate role's view, one-launch form only)
        else if (mode == 1) hipLaunchKernelGGL((k_mid<2, 1>), dim3(grid), dim3(512), 0, s, d.ctl, d, a, d);
        else hipLaunchKernelGGL((k_mid<2, 2>), dim3(grid), dim3(512), 0, s, d.ctl, d, a, d);
    } else {
        if (mode == 0) hipLaunchKernelGGL((k_mid<4, 0>), dim3(grid), dim3(512), 0, s, d.ctl, d, a, d);
        else if (mode == 1) hipLaunchKernelGGL((k_mid<4, 1>), dim3(grid), dim3(512), 0, s, d.ctl, d, a, d);
        else hipLaunchKernelGGL((k_mid<4, 2>), dim3(grid), dim3(512), 0, s, d.ctl, d, a, d);
    }
}
template <int KC> static void launch_downdate2(const RekfDev &d, int grid, hipStream_t s, bool first_on_device, const RekfDev *dn, const RekfFrontArgs *an, int n_front)
{
    constexpr int BYTES = 4 * KC * 64 * (int)sizeof(double) + (KC < 64 ? 16384 : 0);
    if (first_on_device) {
        (void)hipFuncSetAttribute((const void *)k_downdate2<KC>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
        (void)hipFuncSetAttribute((const void *)k_dd_front<KC>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
    }
    if (dn) hipLaunchKernelGGL((k_dd_front<KC>), dim3(grid + n_front), dim3(256), BYTES, s, d, *dn, *an);
    else hipLaunchKernelGGL((k_downdate2<KC>), dim3(grid), dim3(256), BYTES, s, d);
}
// host half of the tile schedule (tests/test_downdate_schedule_cpu.py restates it): T class-A workgroups (diagonal tile + the
// one below) + equal ranges of the rest
static void downdate_schedule(int n_ub, int slots, int &grid, int &dd_lo, int &dd_x, int &dd_sub)
{
    const int T = (rekf_strip_base(n_ub) >= 0) ? n_ub / DT : (n_ub + DT - 1) / DT;      // the kernel's own T: a thin border rides on the diagonal tiles as strips
    const int room = (slots - T > 1) ? slots - T : 1;
    int nB = (T - 1) * (T - 2) / 2;
    dd_sub = 2;
    if ((nB + room - 1) / room < 3) { dd_sub = 1; nB = T * (T - 1) / 2; }   // small states: a class-A workgroup keeps to its diagonal tile
    dd_lo = nB / room; dd_x = nB - dd_lo * room;          // the first dd_x class-B workgroups take dd_lo + 1 tiles, the others dd_lo
    grid = T + (dd_lo > 0 ? room : dd_x);
    if (grid >= 64) grid = (grid + 7) & ~7;         // multiple of 8 for the per-XCD numbering (workgroups past the last range return at once)
}
// per-DEVICE facts the launches below need: the CU count, and which kernel variants have their opt-in to more than 64 KiB of dynamic LDS
constexpr int DD_MAX_DEV = 64;
static struct {
    int n_cu_of[DD_MAX_DEV] = {0};
    unsigned attr_done[DD_MAX_DEV] = {0};     // bit KC/16 : k_downdate2<KC> / k_dd_front<KC>
    unsigned attr_one[DD_MAX_DEV] = {0};      // bit 2 (KC/16 - 1) + (MODE - 2) : k_mid<2, MODE, KC>, the one-launch form
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
        g_dd_cache.attr_one[slot] = 0;
    }
    return slot;
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
// ONE launch per scan for small states (k_mid<2, MODE, KC>): the previous scan's downdate (dd: its device view, the panels of ITS k_mid), this
// scan's front end (one observation per workgroup) and its mid role.  Returns the number of downdate workgroups (the host's share of the
// RekfCtl::dd_done bookkeeping).
template <int MODE, int KC> static void launch_one(const RekfDev &dp, const RekfDev &d, const RekfFrontArgs &a, int grid, bool first_on_device, hipStream_t s)
{
    constexpr int BYTES = 4 * KC * 64 * (int)sizeof(double) + 16384;
    if (first_on_device) (void)hipFuncSetAttribute((const void *)k_mid<2, MODE, KC>, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
    hipLaunchKernelGGL((k_mid<2, MODE, KC>), dim3(grid), dim3(512), BYTES, s, d.ctl, d, a, dp);
}
// does a state of at most n_ub rows (its held-back downdate: dd_n_ub) with K observations have the one-launch form on this device?
// (every workgroup of that launch holds a CU: all of them must be resident together)
int rekf_one_launch_fits(int dd_n_ub, int n_ub, int K)
{
    std::lock_guard<std::mutex> guard(g_dd_cache.mu);
    int dev = 0;
    const int slot = dd_cache_slot(dev);
    const int slots = g_dd_cache.n_cu_of[slot] - K - (n_ub + MID_ROWS - 1) / MID_ROWS;
    if (K < 1 || K > 32 || slots < 8) return 0;
    int grid_dd, dd_lo, dd_x, dd_sub;
    downdate_schedule(dd_n_ub, slots, grid_dd, dd_lo, dd_x, dd_sub);
    return grid_dd <= slots ? 1 : 0;
}
int rekf_launch_one(const RekfDev &dd, int dd_n_ub, const RekfDev &d, RekfFrontArgs &a, int n_ub, int m_ub, bool mode_grow, unsigned dd_done_before, hipStream_t s)
{
    const int kc = (dd.kc_ub < 16) ? 16 : dd.kc_ub;
    if (m_ub > 32 || kc > 32 || a.K < 1 || a.K > 32) return 0;
    std::lock_guard<std::mutex> guard(g_dd_cache.mu);
    int dev = 0;
    const int slot = dd_cache_slot(dev);
    const int n_mid = (n_ub + MID_ROWS - 1) / MID_ROWS;
    int slots = g_dd_cache.n_cu_of[slot] - a.K - n_mid;              // one workgroup per CU (the launch's LDS), everybody resident at once
    if (slots < 8) return 0;
    int grid_dd, dd_lo, dd_x, dd_sub;
    downdate_schedule(dd_n_ub, slots, grid_dd, dd_lo, dd_x, dd_sub);
    if (grid_dd > slots) return 0;
    RekfDev dp = dd;
    dp.dd_lo = dd_lo; dp.dd_x = dd_x; dp.dd_sub = dd_sub; dp.dd_grid = grid_dd;
    if (dd.n_known < 0) { dp.dd_lo = 0; dp.dd_x = 0; dp.dd_sub = 0; }   // (a bound only: the kernel derives the schedule from the real n)
    a.dd_in_mid = grid_dd;
    a.dd_target = dd_done_before + (unsigned)grid_dd;
    a.front_in_mid = a.K;
    const int mode = mode_grow ? 3 : 2;
    const unsigned bit = 1u << (2 * (kc / 16 - 1) + (mode - 2));
    const bool first = !(g_dd_cache.attr_one[slot] & bit) || dev != slot;
    g_dd_cache.attr_one[slot] |= bit;
    const int grid = grid_dd + a.K + n_mid;
    if (kc == 32) { if (mode == 3) launch_one<3, 32>(dp, d, a, grid, first, s); else launch_one<2, 32>(dp, d, a, grid, first, s); }
    else { if (mode == 3) launch_one<3, 16>(dp, d, a, grid, first, s); else launch_one<2, 16>(dp, d, a, grid, first, s); }
    return grid_dd;
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
