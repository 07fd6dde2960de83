// det2d.hip -- MI355X-native 2D reflector detector behind the C ABI of include/rdet.h.
//
// Replaces reflector_detect::LaserReflectorDetect::HandleLaserScan (reference
// src/reflector_detect/laser/laser_reflector_detect.cc:23-316) and the PoseExtrapolator it
// owns (src/reflector_detect/laser/pose_extrapolator.cc).
//
// The reference is a sequential state machine over the beams.  Its decisions are, however,
// local: with prev(i) = the previous bright beam, a bright beam i
//     continues the run   if i - prev == 1                                         (:96-101)
//     bridges a gap       if i - prev < 4, |r_i - r_prev| < 0.3, beam i+1 bright   (:111-138)
//     closes the run and starts a new one otherwise                                (:140-169)
// so one 1024-thread workgroup resolves the runs of a whole scan in LDS with five block-wide scans
// (valid count, last valid point, previous bright beam, run index, accepted-run index), gates all
// closed runs in parallel (:147-156) and resolves the first/last-run wrap logic on one lane
// (:178-236), while N/256 other workgroups of the SAME launch de-skew every point (:239-270) and
// compute what every candidate beam would add to a centre; workgroup 0 then averages each accepted
// cluster in beam order with float32 running sums (:277-306), one wave per cluster.
//
// I/O per scan: the host writes ranges/intensities straight into fine-grained device memory (PCIe
// BAR, posted writes), launches ONE kernel, and polls a tagged 16-byte slot in pinned host memory
// that the kernel stores the centres into -- no copy engine, no completion-signal wait.
//
// cos/sin of the float32-accumulated beam angle (Q15, :51,:175) come from a table the host
// fills with the same libm the reference would call, as does the odometry yaw 2 atan2(q.z, q.w),
// so run membership and gating see the same float32 points as a CPU build.  Everything is float32
// where the reference is (geometry, point time stored in a Vector3f) and FP64 where it is (poses,
// odometry).
#include "../../include/rdet.h"
#include "host_visible.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {

// PoseExtrapolator sample: nav_msgs::Odometry reduced to what pose_extrapolator.cc reads; yaw = 2 atan2(q.z, q.w)
// (:36,:53) is taken once on the host with the libm a CPU build would call
struct Odom {
    double time, px, py, yaw, vx, vy, wz;
};

struct R2d { double x, y, a; };
struct R2f { float x, y, a; };

struct Det2dArgs {
    // options (laser_reflector_detect.h:8-15)
    double intensity_min, min_length, length_error;
    float opt_range_min, opt_range_max;
    // message header
    float msg_range_min, msg_range_max, angle_increment;
    double first_point_time, point_delta_t;
    int N, is_circle, max_centers;
    int seq;                       // call number: written to Det2dOut::seq when the centres are in host memory
    unsigned done_target;          // value the monotonic counter of published per-beam workgroups reaches with THIS scan (the counter is
                                   // never reset by the kernel: a straggler of a scan that timed out cannot satisfy the next scan's wait)
    // sensor_to_base_link as Rigid2f + host-evaluated cos/sin of its angle
    float s2b_x, s2b_y, s2b_a, s2b_c, s2b_s;
    // pose extrapolator state: 0, 1 or 2 samples (front, back)
    int n_odom;
    Odom front, back;
};

// What the kernel hands back, in pinned host memory.  Every slot is ONE 16-byte system-scope store that carries the
// scan's number, so the kernel never waits for a PCIe write to be acknowledged and the host never needs an ordering
// between slots: it polls head.seq, then takes each centre once that centre's own tag has arrived.
struct Det2dSlot { float x, y; int seq, pad; };
struct Det2dHead { int K, n_returns, runs_err, seq; };      // runs_err = n_runs | (-err) << 16
struct Det2dOut {
    Det2dHead head;
    Det2dSlot centers[RDET_MAX_CENTERS];
    int fin, pad[3];               // = Det2dArgs::seq once the kernel has nothing left to write
};

#define RDET2D_MAX_BEAMS 8192             // workgroup 0 holds a whole scan in LDS (17 B per beam)
#define RDET2D_GROUP 256                  // beams per per-beam workgroup

struct Det2dBufs {
    const float *ranges, *intens;      // pinned host memory, read in place (the scan never takes a copy engine)
    const float *ang, *cosv, *sinv;    // device: the beam-angle table
    // per-beam results, written by workgroups 1.. and read by workgroup 0 of the same launch: agent-scope atomics only
    unsigned long long *contrib;       // float2 bits: the point a beam would add to its cluster's centre
    unsigned long long *cmask;         // per 64 beams: which of them have a contribution
    unsigned long long *returns_all;   // float2 bits: de-skewed return of beam j (valid beams)
    float2 *returns;                   // de-skewed point cloud in point_cloud order (GetRangeData)
    Det2dOut *out;                     // pinned host memory, written in place
    int *done;                         // device: per-beam workgroups that have published
#ifdef RDET_DEBUG_MARKS
    unsigned long long *marks;         // pinned host: [0..15] workgroup 0, [16..31] workgroup 1 (shader clock)
#endif
};
#ifdef RDET_DEBUG_MARKS
#define DMARK(slot) do { if (threadIdx.x == 0) B.marks[slot] = __builtin_readcyclecounter(); } while (0)
#else
#define DMARK(slot) do { } while (0)
#endif

// ---- inter-workgroup traffic: write-through / cache-bypassing accesses instead of fences (an agent-scope release
// fence writes the whole L2 back: microseconds on a kernel that lives for ten) ----------------------------------
__device__ static void publish_u64(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ static unsigned long long fetch_u64(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ static unsigned long long f2_bits(float2 v)
{
    return (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32);
}
__device__ static float2 bits_f2(unsigned long long b)
{
    return make_float2(__uint_as_float((unsigned)b), __uint_as_float((unsigned)(b >> 32)));
}
__device__ static void stores_landed() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// one 16-byte write-through store to host memory (a single PCIe write: the tag in .w lands with the payload)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ static void host_store16(void *p, unsigned a, unsigned b, unsigned c, unsigned d)
{
    const u32x4 v = {a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// ---- Rigid2 algebra (rigid_transform.h:46-51,62-67,87-102), no FMA contraction -------------
// inverse of r, given c = cos(r.a), s = sin(r.a): cos(-a) = c and sin(-a) = -s exactly
__device__ static R2d r2_inverse_cs(double c, double s, R2d r)
{
#pragma clang fp contract(off)
    R2d o;
    o.a = -r.a;
    o.x = -(c * r.x + s * r.y);
    o.y = -((-s) * r.x + c * r.y);
    return o;
}
__device__ static R2d r2_mul_cs(double lc, double ls, R2d l, R2d r)   // l with cos/sin(l.a) given
{
#pragma clang fp contract(off)
    R2d o;
    o.x = (lc * r.x + (-ls) * r.y) + l.x;
    o.y = (ls * r.x + lc * r.y) + l.y;
    o.a = l.a + r.a;
    return o;
}
__device__ static R2f r2_cast(R2d r)
{
    R2f o; o.x = (float)r.x; o.y = (float)r.y; o.a = (float)r.a; return o;
}
#include "glibc_sincosf.h"     // float32 sin / cos with the host libm's bits (the reference calls std::sin / std::cos on floats)

__device__ static float2 r2f_apply_cs(float c, float s, float tx, float ty, float px, float py)
{
#pragma clang fp contract(off)
    float2 o;
    o.x = (c * px + (-s) * py) + tx;
    o.y = (s * px + c * py) + ty;
    return o;
}
__device__ static float2 r2f_apply(R2f r, float px, float py)
{
    float s, c;
    glibc_sincosf(r.a, &s, &c);
    return r2f_apply_cs(c, s, r.x, r.y, px, py);
}

// ---- PoseExtrapolator (pose_extrapolator.cc:34-84,102-129) ------------------------------------
// One straight-line evaluation for both branches of Interpolator (sample before / after `time`): the branches differ
// in the sign of the velocity terms only, and a - b == a + (-b) exactly, so lanes of a wave never split around the
// FP64 sincos.  *c_out / *s_out = cos / sin of the returned angle.
__device__ static R2d extrapolator_pose(const Det2dArgs &A, double time, double *c_out, double *s_out)
{
#pragma clang fp contract(off)
    R2d o = {0, 0, 0};
    *c_out = 1.0; *s_out = 0.0;
    if (A.n_odom == 0) return o;
    // time <= front: the first sample; t >= back, or in between: always the LAST sample (:76-82, Q14)
    const bool uf = time <= A.front.time;
    const double st_time = uf ? A.front.time : A.back.time, px = uf ? A.front.px : A.back.px, py = uf ? A.front.py : A.back.py;
    const double yaw = uf ? A.front.yaw : A.back.yaw, vx = uf ? A.front.vx : A.back.vx, vy = uf ? A.front.vy : A.back.vy;
    const double wz = uf ? A.front.wz : A.back.wz;
    const bool past = st_time <= time;                               // :36-50, else :51-66
    const double delta_t = past ? st_time - time : time - st_time;
    const double now_yaw = yaw - wz * delta_t;                       // sign as in the reference for both (Q14)
    double s, c;
    sincos(now_yaw, &s, &c);
    const double sg = past ? -1.0 : 1.0;
    const double ax = vx * delta_t, ay = vy * delta_t;
    o.x = (px + sg * (ax * c)) - sg * (ay * s);
    o.y = (py + sg * (ax * s)) + sg * (ay * c);
    o.a = now_yaw;
    *c_out = c; *s_out = s;
    return o;
}

// ---- block-wide exclusive scans over 1024 per-thread values -----------------------------------
// in-wave inclusive scan by six DPP steps (row shifts 1/2/4/8, then row_bcast:15 / row_bcast:31 across the rows of 16),
// the 16 wave totals through an LDS slot that each scan of the kernel uses once: ONE barrier per scan
struct ScanSum { static constexpr int id = 0; __device__ static int f(int a, int b) { return a + b; } };
struct ScanMax { static constexpr int id = -1; __device__ static int f(int a, int b) { return max(a, b); } };   // values >= -1
template <class Op, int CTRL, int ROW_MASK> __device__ static int dpp_step(int v)
{
    return Op::f(v, __builtin_amdgcn_update_dpp(Op::id, v, CTRL, ROW_MASK, 0xf, false));
}
template <class Op> __device__ static int wave_incl_scan(int v)
{
    v = dpp_step<Op, 0x111, 0xf>(v);        // row_shr:1
    v = dpp_step<Op, 0x112, 0xf>(v);        // row_shr:2
    v = dpp_step<Op, 0x114, 0xf>(v);        // row_shr:4
    v = dpp_step<Op, 0x118, 0xf>(v);        // row_shr:8
    v = dpp_step<Op, 0x142, 0xa>(v);        // row_bcast:15 into rows 1 and 3
    v = dpp_step<Op, 0x143, 0xc>(v);        // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ static int block_excl_sum(int v, int *lds, int *total)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int incl = wave_incl_scan<ScanSum>(v);
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int c = lds[w]; if (w < wave) base += c; tot += c; }
    if (total) *total = tot;
    return base + incl - v;
}
__device__ static int block_excl_max(int v, int *lds, int *total)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int incl = wave_incl_scan<ScanMax>(v);
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    int base = -1, tot = -1;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int c = lds[w]; if (w < wave) base = max(base, c); tot = max(tot, c); }
    if (total) *total = tot;
    const int prev = __builtin_amdgcn_update_dpp(-1, incl, 0x138, 0xf, 0xf, false);   // wave_shr:1: inclusive value of the previous lane
    return max(base, prev);
}

// ================================================================================================
// One launch per scan.  Workgroup 0 is the run state machine: beams -> runs -> gated clusters, entirely in LDS.
// Workgroups 1.. own 256 beams each and do everything that does not depend on the runs: the de-skewed return of
// every valid beam (:246-258) and, for every beam that COULD belong to a cluster (a bright beam, or a finite beam
// with a bright beam at most three ahead -- the only beams a bridged gap can hold, :111), the point it would add
// to its cluster's centre (:277-299).  That is all of the FP64 trigonometry, spread over N / 256 CUs.  They publish
// through `done`; workgroup 0 waits for them once its clusters are known and takes the ordered float32 sums
// (:300-305), one wave per cluster.  The dependency is one-way (nobody waits for workgroup 0): no deadlock.
// ================================================================================================

// ---- workgroups 1..: per-beam work.  Waves 0-3 own the beams, wave 4 the halo, wave 5 finds the last valid beam
// and inverts the scan-end pose, wave 6 finds the first valid beam; the rest only keep the barriers company.
__device__ static void det2d_beams(const Det2dArgs &A, const Det2dBufs &B, const int blk)
{
    __shared__ unsigned char s_bf[RDET2D_GROUP + 8];
    __shared__ double s_inv[5];
    __shared__ float s_tb[4];
    __shared__ int s_edge[2];                       // first / last valid beam of the scan
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, N = A.N;
    const int j = blk * RDET2D_GROUP + tid, jc = min(j, N - 1);
    const bool owner = tid < RDET2D_GROUP;
    if (blk == 0) DMARK(16);

    float rgj = -1.f, itj = 0.f, cvj = 0.f, svj = 0.f, ang_ahead[3] = {0.f, 0.f, 0.f};
    if (tid < RDET2D_GROUP + 4 && j < N) { rgj = B.ranges[j]; itj = B.intens[j]; }      // (-1 marks "no beam": range_min >= 0)
    if (owner) {
        cvj = B.cosv[jc]; svj = B.sinv[jc];
#pragma unroll
        for (int d = 0; d < 3; ++d) ang_ahead[d] = B.ang[min(j + 1 + d, N - 1)];
    }
    if (wave == 5) {                                 // last valid beam: 64 beams at a time from the end
        int found = -1;
        for (int base = N - 1; base >= 0 && found < 0; base -= 64) {
            const int i = base - lane;
            const float r = i >= 0 ? B.ranges[i] : -1.f;
            const unsigned long long m = __ballot(r >= A.msg_range_min && r <= A.msg_range_max);
            if (m) found = base - (__ffsll((long long)m) - 1);
        }
        if (lane == 0) s_edge[1] = found;
    }
    if (wave == 6) {                                 // first valid beam
        int found = 0x7fffffff;
        for (int base = 0; base < N && found == 0x7fffffff; base += 64) {
            const int i = base + lane;
            const float r = i < N ? B.ranges[i] : -1.f;
            const unsigned long long m = __ballot(r >= A.msg_range_min && r <= A.msg_range_max);
            if (m) found = base + (__ffsll((long long)m) - 1);
        }
        if (lane == 0) s_edge[0] = found;
    }
    const bool valid = owner && rgj >= A.msg_range_min && rgj <= A.msg_range_max;                                  // :65
    const bool cand = j < N && A.opt_range_min <= rgj && rgj <= A.opt_range_max && (double)itj > A.intensity_min;   // :73-75
    if (tid < RDET2D_GROUP + 4) s_bf[tid] = cand ? 4 : 0;
    if (blk == 0) DMARK(17);
    __syncthreads();
    if (blk == 0) DMARK(18);
    const int first_valid_all = s_edge[0], last_valid_all = s_edge[1];
    if (wave == 5 && last_valid_all >= 0) {          // scan-end pose (:252-253, :299) and its inverse
        double c, s;
        const R2d mtp = extrapolator_pose(A, (double)(float)(A.first_point_time + last_valid_all * A.point_delta_t), &c, &s);
        const R2d inv = r2_inverse_cs(c, s, mtp);
        const R2f tb = r2_cast(inv);
        float tc, ts;
        glibc_sincosf(tb.a, &ts, &tc);
        if (lane == 0) {
            s_inv[0] = inv.x; s_inv[1] = inv.y; s_inv[2] = inv.a; s_inv[3] = c; s_inv[4] = -s;
            s_tb[0] = tb.x; s_tb[1] = tb.y; s_tb[2] = tc; s_tb[3] = ts;
        }
    }
    const bool bright = owner && cand && first_valid_all <= j;                      // "a point exists" guard (:77)
    // a gap beam (:115-130) is re-projected from the NEXT bright beam's accumulated angle
    int ahead = 0;
    if (owner && !bright && j < N && !isinf(rgj)) {
#pragma unroll
        for (int d = 3; d >= 1; --d)
            if ((s_bf[tid + d] & 4) && first_valid_all <= j + d) ahead = d;
    }
    const float tj = (float)(A.first_point_time + j * A.point_delta_t);             // :66 (stored in a Vector3f)
    R2d pose_j = {0, 0, 0};
    double pc = 1.0, ps = 0.0;
    if (wave < RDET2D_GROUP / 64 && last_valid_all >= 0) pose_j = extrapolator_pose(A, (double)tj, &pc, &ps);
    float2 pt_j = make_float2(0.f, 0.f);
    if (valid) {
#pragma clang fp contract(off)
        pt_j = r2f_apply_cs(A.s2b_c, A.s2b_s, A.s2b_x, A.s2b_y, rgj * cvj, rgj * svj);               // :68-70
    }
    if (blk == 0) DMARK(19);
    __syncthreads();
    if (blk == 0) DMARK(20);
    bool has = false;
    if (owner && last_valid_all >= 0) {
        if (valid) {                                                                // de-skew (:246-258)
            const R2d inv = {s_inv[0], s_inv[1], s_inv[2]};
            const R2f rel = r2_cast(r2_mul_cs(s_inv[3], s_inv[4], inv, pose_j));
            publish_u64(B.returns_all + j, f2_bits(r2f_apply(rel, pt_j.x, pt_j.y)));
        }
        float2 p = pt_j;
        R2d pose_p = pose_j;
        if (bright) {
            has = true;
            if (!valid) {                       // bright beyond the message's own range limits: point_cloud.back() (:87)
#pragma clang fp contract(off)
                int lv = j - 1;
                float rl = 0.f;
                while (lv >= 0) { rl = B.ranges[lv]; if (rl >= A.msg_range_min && rl <= A.msg_range_max) break; --lv; }
                p = r2f_apply_cs(A.s2b_c, A.s2b_s, A.s2b_x, A.s2b_y, rl * B.cosv[lv], rl * B.sinv[lv]);
                double c2, s2;
                pose_p = extrapolator_pose(A, (double)(float)(A.first_point_time + lv * A.point_delta_t), &c2, &s2);
            }
        } else if (ahead) {
#pragma clang fp contract(off)
            has = true;
            const float a_next = ahead == 1 ? ang_ahead[0] : ahead == 2 ? ang_ahead[1] : ang_ahead[2];
            const float angle_gap = a_next - A.angle_increment * (float)ahead;      // :117
            float gs, gc;
            glibc_sincosf(angle_gap, &gs, &gc);
            p = r2f_apply_cs(A.s2b_c, A.s2b_s, A.s2b_x, A.s2b_y, rgj * gc, rgj * gs);
        }
        if (has) {
            const R2f pose = r2_cast(pose_p);                                       // :287,:293
            const float2 po = r2f_apply(pose, p.x, p.y);
            publish_u64(B.contrib + j, f2_bits(r2f_apply_cs(s_tb[2], s_tb[3], s_tb[0], s_tb[1], po.x, po.y)));
        }
    }
    if (owner) {
        const unsigned long long hm = __ballot(has);
        if (lane == 0 && j < N) publish_u64(B.cmask + (j >> 6), hm);
    }
    stores_landed();
    if (blk == 0) DMARK(21);
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(B.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blk == 0) DMARK(22);
}

// ---- workgroup 0: runs, gates, clusters, sums --------------------------------------------------
__device__ static void det2d_runs(const Det2dArgs &A, const Det2dBufs &B, const int n_beam_groups)
{
    __shared__ float s_rg[RDET2D_MAX_BEAMS];
    __shared__ float2 s_pt[RDET2D_MAX_BEAMS];          // point in base_link (valid beams)
    __shared__ short s_lv[RDET2D_MAX_BEAMS];           // last valid beam <= i (point_cloud.back() at beam i), -1 if none
    __shared__ unsigned char s_fl[RDET2D_MAX_BEAMS];   // 1 valid, 2 bright (after the "a point exists" guard), 4 bright before it, 16 intensity above the gate
    __shared__ short s_rf[RDET2D_MAX_BEAMS / 2 + 4], s_rl[RDET2D_MAX_BEAMS / 2 + 4];   // first / last beam of run r
    __shared__ int lds_scan[5][16];                    // one slot per block scan (no barrier to recycle it)
    __shared__ int s_tot[4];
    __shared__ int s_cl[4 * RDET_MAX_CENTERS + 8];     // cluster segments: first0,last0,first1,last1
    const int tid = threadIdx.x;
    const int N = A.N;
    const int CH = (N + 1023) / 1024;
    const int b0 = tid * CH, b1 = min(N, b0 + CH);
    constexpr int MAXCH = RDET2D_MAX_BEAMS / 1024;
    DMARK(0);

    // ---- pass 1: points, validity, brightness (:63-83), beam i = tid + 1024 q so that a wave reads whole lines
    // (the scan is in host memory: every load instruction is a PCIe read of its own)
    {
        float rr[MAXCH], ii[MAXCH], cc[MAXCH], ss[MAXCH];
#pragma unroll
        for (int q = 0; q < MAXCH; ++q) {
            const int i = tid + 1024 * q;
            const bool in = i < N;
            rr[q] = in ? B.ranges[i] : 0.f; ii[q] = in ? B.intens[i] : 0.f; cc[q] = in ? B.cosv[i] : 0.f; ss[q] = in ? B.sinv[i] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < MAXCH; ++q) {
#pragma clang fp contract(off)
            const int i = tid + 1024 * q;
            if (i >= N) continue;
            const float range = rr[q];
            unsigned char f = 0;
            s_rg[i] = range;
            if (range >= A.msg_range_min && range <= A.msg_range_max) {
                f |= 1;
                s_pt[i] = r2f_apply_cs(A.s2b_c, A.s2b_s, A.s2b_x, A.s2b_y, range * cc[q], range * ss[q]);
            }
            if ((double)ii[q] > A.intensity_min) {
                f |= 16;
                if (A.opt_range_min <= range && range <= A.opt_range_max) f |= 4;
            }
            s_fl[i] = f;
        }
    }
    __syncthreads();
    DMARK(1);
    // from here a thread owns CH <= 8 consecutive beams
    unsigned char fl[MAXCH];
    int prevb_r[MAXCH];
    int cnt_valid = 0, last_valid = -1;
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) {
        const int i = b0 + q;
        fl[q] = 0;
        if (q >= CH || i >= b1) continue;
        fl[q] = s_fl[i];
        if (fl[q] & 1) { ++cnt_valid; last_valid = i; }
    }
    int n_cloud;
    const int cloud_base = block_excl_sum(cnt_valid, lds_scan[0], &n_cloud);
    int lv = block_excl_max(last_valid, lds_scan[1], nullptr);
    DMARK(2);
    // ---- pass 2: point_cloud.back(), guarded bright flag, previous bright beam
    int last_bright = -1;
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) {
        const int i = b0 + q;
        if (q >= CH || i >= b1) continue;
        unsigned char f = fl[q];
        if (f & 1) lv = i;
        s_lv[i] = (short)lv;
        if ((f & 4) && lv >= 0) { f |= 2; last_bright = i; }
        fl[q] = f;
    }
    int last_bright_all;
    int pb = block_excl_max(last_bright, lds_scan[2], &last_bright_all);
    DMARK(3);
    // ---- pass 3: run starts (:85-169) and run index
    int n_start = 0;
    unsigned start_mask = 0;
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) {
        const int i = b0 + q;
        prevb_r[q] = -1;
        if (q >= CH || i >= b1 || !(fl[q] & 2)) continue;
        prevb_r[q] = pb;
        bool start = pb < 0;
        if (!start && i - pb != 1) {
            const int nx = (i + 1 < N) ? i + 1 : i;
            const bool gap = (i - pb < 4) && (fabs((double)(s_rg[i] - s_rg[pb])) < 0.3) && (s_fl[nx] & 16);   // :111
            start = !gap;
        }
        if (start) start_mask |= 1u << q;
        n_start += start ? 1 : 0;
        pb = i;
    }
    int n_runs;
    int rid = block_excl_sum(n_start, lds_scan[3], &n_runs);
    DMARK(4);
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) {
        const int i = b0 + q;
        if (q >= CH || i >= b1 || !(fl[q] & 2)) continue;
        if (start_mask & (1u << q)) {      // start of run `rid`
            s_rf[rid] = (short)i;
            if (prevb_r[q] >= 0) s_rl[rid - 1] = (short)prevb_r[q];
            ++rid;
        }
    }
    if (tid == 0 && n_runs > 0) s_rl[n_runs - 1] = (short)last_bright_all;
    __syncthreads();
    DMARK(5);

    // ---- gate the closed runs (:147-156), compact the accepted ones
    const int n_closed = (n_runs > 0) ? n_runs - 1 : 0;
    const int RCH = (n_closed + 1023) / 1024;
    const int r0 = tid * RCH, r1 = min(n_closed, r0 + RCH);
    int n_acc_local = 0;
    unsigned acc_mask = 0;
    for (int r = r0; r < r1; ++r) {
#pragma clang fp contract(off)
        const int fi = s_rf[r], li = s_rl[r];
        const float2 pf = s_pt[s_lv[fi]], pl = s_pt[s_lv[li]];
        const float len = hypotf(pf.x - pl.x, pf.y - pl.y);
        const bool ok = (A.is_circle && fi == 0) || (fabs((double)len - A.min_length) < A.length_error);
        if (ok) { acc_mask |= 1u << (r - r0); ++n_acc_local; }
    }
    int n_acc;
    int cidx = block_excl_sum(n_acc_local, lds_scan[4], &n_acc);
    DMARK(6);
    for (int r = r0; r < r1; ++r) {
        if (!(acc_mask & (1u << (r - r0)))) continue;
        if (cidx < RDET_MAX_CENTERS) {
            s_cl[4 * cidx + 0] = s_rf[r]; s_cl[4 * cidx + 1] = s_rl[r];
            s_cl[4 * cidx + 2] = -1; s_cl[4 * cidx + 3] = -1;
        }
        ++cidx;
    }
    __syncthreads();

    // ---- last / first reflector (:178-236), one lane
    if (tid == 0) {
#pragma clang fp contract(off)
        int n_cl = n_acc, off = 0, err = 0;
        if (n_acc > RDET_MAX_CENTERS) { err = RDET_ERR_CAPACITY; n_cl = RDET_MAX_CENTERS; }
        if (n_runs > 0 && !err) {
            const int Lr = n_runs - 1;
            const int lf = s_rf[Lr], ll = s_rl[Lr];
            const float2 last_first_pt = s_pt[s_lv[lf]], last_pt = s_pt[s_lv[ll]];
            const float len = hypotf(last_first_pt.x - last_pt.x, last_first_pt.y - last_pt.y);
            const bool len_ok = fabs((double)len - A.min_length) < A.length_error;
            if (n_cl > 0) {
                const int first_id = s_cl[0];
                const float2 first_pt = s_pt[s_lv[s_cl[0]]];
                const float2 first_last_pt = s_pt[s_lv[s_cl[1]]];
                const float dx = last_pt.x - first_pt.x, dy = last_pt.y - first_pt.y;
                if (A.is_circle && first_id == 0 && ll == N - 1 && sqrtf(dx * dx + dy * dy) < 0.1) {   // :188-195
                    s_cl[2] = lf; s_cl[3] = ll;
                } else if (len_ok) {                                                                   // :196-204
                    if (n_cl < RDET_MAX_CENTERS) {
                        s_cl[4 * n_cl + 0] = lf; s_cl[4 * n_cl + 1] = ll; s_cl[4 * n_cl + 2] = -1; s_cl[4 * n_cl + 3] = -1;
                        ++n_cl;
                    } else err = RDET_ERR_CAPACITY;
                }
                if (A.is_circle && ll == 0) {                                                          // :205-214
                    const float fx = first_last_pt.x - last_first_pt.x, fy = first_last_pt.y - last_first_pt.y;
                    if (fabs((double)sqrtf(fx * fx + fy * fy) - A.min_length) >= A.length_error) off = 1;
                }
            } else if (len_ok) {                                                                        // :216-224
                s_cl[0] = lf; s_cl[1] = ll; s_cl[2] = -1; s_cl[3] = -1;
                n_cl = 1;
            }
        }
        // (no bright beam at all: the reference touches an empty deque, :226 -- defined as no reflectors)
        int K = n_cl - off;
        if (K < 0) K = 0;
        if (K > A.max_centers) { err = RDET_ERR_BUFFER; K = 0; }
        if (n_cloud == 0) K = 0;
        DMARK(7);
        // the per-beam workgroups' results (bounded wait: a lost workgroup becomes an error, not a hang)
        const unsigned long long t0 = __builtin_readcyclecounter();
        while ((int)((unsigned)__hip_atomic_load(B.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - A.done_target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_readcyclecounter() - t0 > 250000000ull) { err = RDET_ERR_HIP; K = 0; break; }   // ~0.1 s
        }
        s_tot[0] = K; s_tot[1] = off; s_tot[2] = err;
        DMARK(8);
    }
    __syncthreads();
    const int K = s_tot[0], off = s_tot[1];
    DMARK(9);
    // ---- per cluster: the float32 running sum in beam order that the reference takes (:300-305), one wave per cluster.
    // A wave first requests the leading 64 beams of up to four of its clusters (one memory round trip for a scan of
    // <= 64 reflectors), then adds them up; longer clusters and the wrapped segment fetch on demand.
    {
#pragma clang fp contract(off)
        const int lane = tid & 63, wave = tid >> 6;
        for (int c0 = wave; c0 < K; c0 += 64) {
            unsigned long long cb0[4], mw0[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + 16 * u;
                cb0[u] = 0; mw0[u] = 0;
                if (c < K) {
                    const int j = min(s_cl[4 * (c + off)] + lane, N - 1);
                    cb0[u] = fetch_u64(B.contrib + j);
                    mw0[u] = fetch_u64(B.cmask + (j >> 6));
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + 16 * u;
                if (c >= K) continue;
                const int k = c + off;
                float cx = 0.f, cy = 0.f;
                int count = 0;
                for (int seg = 0; seg < 2; ++seg) {
                    const int fi = s_cl[4 * k + 2 * seg], li = s_cl[4 * k + 2 * seg + 1];
                    if (fi < 0) continue;
                    for (int j0 = fi; j0 <= li; j0 += 64) {
                        const int j = min(j0 + lane, N - 1);
                        const bool pre = seg == 0 && j0 == fi;
                        const unsigned long long cb = pre ? cb0[u] : fetch_u64(B.contrib + j);
                        const unsigned long long mw = pre ? mw0[u] : fetch_u64(B.cmask + (j >> 6));
                        const bool mem = j0 + lane <= li && ((mw >> (j & 63)) & 1ull);
                        const float2 v = mem ? bits_f2(cb) : make_float2(0.f, 0.f);
                        unsigned long long mask = __ballot(mem);
                        count += __popcll(mask);
                        while (mask) {                      // (the ballot is wave-uniform: scalar loop, v_readlane)
                            const int b = __builtin_amdgcn_readfirstlane(__ffsll((long long)mask) - 1);
                            mask &= mask - 1;
                            cx += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.x), b));
                            cy += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.y), b));
                        }
                    }
                }
                if (lane == 0)                                                       // :305, straight into host memory
                    host_store16(B.out->centers + c, __float_as_uint(cx / (float)count), __float_as_uint(cy / (float)count), (unsigned)A.seq, 0u);
            }
        }
    }
    DMARK(10);
    if (tid == 0)      // the host polls head.seq: the centres are its to read while this workgroup still compacts the point cloud
        host_store16(&B.out->head, (unsigned)K, (unsigned)n_cloud, (unsigned)n_runs | ((unsigned)(-s_tot[2]) << 16), (unsigned)A.seq);
    // ---- GetRangeData's point cloud: the de-skewed returns in point_cloud order
    {
        unsigned long long rv[MAXCH];
#pragma unroll
        for (int q = 0; q < MAXCH; ++q) {
            const int i = b0 + q;
            rv[q] = (q < CH && i < b1 && (fl[q] & 1) && s_tot[2] != RDET_ERR_HIP) ? fetch_u64(B.returns_all + i) : 0ull;
        }
        int c = cloud_base;
#pragma unroll
        for (int q = 0; q < MAXCH; ++q)
            if (q < CH && b0 + q < b1 && (fl[q] & 1)) B.returns[c++] = bits_f2(rv[q]);
    }
    stores_landed();
    __syncthreads();
    if (tid == 0) {
        stores_landed();
        __hip_atomic_store(&B.out->fin, A.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (the per-beam workgroups were done before the sums)
    }
    DMARK(11);
}

__global__ __launch_bounds__(1024) void k_det2d(Det2dArgs A, Det2dBufs B)
{
    if (blockIdx.x == 0) det2d_runs(A, B, (int)gridDim.x - 1);
    else det2d_beams(A, B, (int)blockIdx.x - 1);
}

}  // namespace

// =================================================================================================
struct rdet2d {
    rdet2d_options opt;
    double s2b[3];
    int max_beams, device;
    hipStream_t stream;
    std::vector<Odom> odom;            // PoseExtrapolator::odometry_data_
    // device buffers
    float *d_ang, *d_cos, *d_sin;      // beam-angle table
    float2 *d_returns;
    unsigned long long *d_returns_all, *d_contrib, *d_cmask;
    int *d_done;
    unsigned done_total = 0;           // per-beam workgroups launched so far (the device counter catches up with it)
    int seq;                           // scans launched; Det2dOut::seq catches up when a scan's centres are in host memory
    // pinned host memory the kernel reads / writes in place
    float *h_scan;                     // ranges | intensities: where the host writes a scan.  Fine-grained DEVICE memory through
                                       // the PCIe BAR when the platform maps it (posted writes, the kernel then reads HBM), else pinned host memory
    bool scan_in_vram;
    Det2dOut *h_out;
    const float *dv_scan;              // the device's view of h_scan / h_out
    Det2dOut *dv_out;
    float *h_table;                    // pinned staging of the beam-angle table: ang | cos | sin
#ifdef RDET_DEBUG_MARKS
    unsigned long long *h_marks;
#endif
    // cached beam-angle table key
    float tab_angle_min, tab_inc;
    int tab_N;
    int last_n_returns;
    std::string hip_error;
};

#define DET_TRY(h, expr)                                                            \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            if (h) (h)->hip_error = std::string(#expr) + ": " + hipGetErrorString(e_); \
            return RDET_ERR_HIP;                                                    \
        }                                                                           \
    } while (0)

extern "C" {

int rdet_abi_version(void) { return RDET_ABI_VERSION; }

const char *rdet_strerror(int code)
{
    switch (code) {
    case RDET_OK: return "ok";
    case RDET_ERR_INVALID: return "invalid argument";
    case RDET_ERR_HIP: return "HIP runtime error";
    case RDET_ERR_BAD_SCAN: return "malformed scan message";
    case RDET_ERR_CAPACITY: return "capacity exceeded";
    case RDET_ERR_BUFFER: return "caller buffer too small";
    default: return "unknown error";
    }
}

int rdet2d_create(const rdet2d_options *opt, const double s2b[3], int max_beams, int device, rdet2d_t **out)
{
    if (!opt || !s2b || !out || max_beams < 1) return RDET_ERR_INVALID;
    *out = nullptr;
    rdet2d_t *h = new (std::nothrow) rdet2d();
    if (!h) return RDET_ERR_INVALID;
    h->opt = *opt;
    std::memcpy(h->s2b, s2b, sizeof(double) * 3);
    h->max_beams = max_beams;
    h->device = device;
    h->tab_N = -1;
    h->last_n_returns = 0;
    const size_t nb = (size_t)max_beams;
    int rc = [&]() -> int {
        DET_TRY(h, hipSetDevice(device));
        DET_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        DET_TRY(h, hipMalloc(&h->d_ang, 4 * nb)); DET_TRY(h, hipMalloc(&h->d_cos, 4 * nb)); DET_TRY(h, hipMalloc(&h->d_sin, 4 * nb));
        DET_TRY(h, hipMalloc(&h->d_returns, 8 * nb)); DET_TRY(h, hipMalloc(&h->d_contrib, 8 * nb));
        DET_TRY(h, hipMalloc(&h->d_returns_all, 8 * nb)); DET_TRY(h, hipMalloc(&h->d_cmask, 8 * (nb / 64 + 1)));
        DET_TRY(h, hipMalloc(&h->d_done, sizeof(int)));
        DET_TRY(h, hipMemset(h->d_done, 0, sizeof(int)));
        h->h_scan = (float *)host_visible::alloc(4 * nb * 2);
        if (h->h_scan) {
            h->scan_in_vram = true;
            h->dv_scan = h->h_scan;
        } else {
            h->scan_in_vram = false;
            DET_TRY(h, hipHostMalloc(&h->h_scan, 4 * nb * 2, hipHostMallocMapped | hipHostMallocCoherent));
            void *dvs = nullptr;
            DET_TRY(h, hipHostGetDevicePointer(&dvs, h->h_scan, 0)); h->dv_scan = (const float *)dvs;
        }
        DET_TRY(h, hipHostMalloc(&h->h_out, sizeof(Det2dOut), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(h->h_out, 0, sizeof(Det2dOut));
        DET_TRY(h, hipHostMalloc(&h->h_table, 4 * nb * 3));
#ifdef RDET_DEBUG_MARKS
        DET_TRY(h, hipHostMalloc(&h->h_marks, 8 * 32, hipHostMallocMapped));
#endif
        void *dv = nullptr;
        DET_TRY(h, hipHostGetDevicePointer(&dv, h->h_out, 0)); h->dv_out = (Det2dOut *)dv;
        return RDET_OK;
    }();
    if (rc != RDET_OK) { std::fprintf(stderr, "rdet2d_create: %s\n", h->hip_error.c_str()); rdet2d_destroy(h); return rc; }
    *out = h;
    return RDET_OK;
}

void rdet2d_destroy(rdet2d_t *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void *ptrs[] = {h->d_ang, h->d_cos, h->d_sin, h->d_returns, h->d_contrib, h->d_returns_all, h->d_cmask, h->d_done};
    for (void *p : ptrs) (void)hipFree(p);
    if (h->h_out) (void)hipHostFree(h->h_out);
    if (h->h_scan) { if (h->scan_in_vram) (void)hipFree(h->h_scan); else (void)hipHostFree(h->h_scan); }
    if (h->h_table) (void)hipHostFree(h->h_table);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int rdet2d_set_sensor_to_base_link(rdet2d_t *h, const double xyyaw[3])
{
    if (!h || !xyyaw) return RDET_ERR_INVALID;
    std::memcpy(h->s2b, xyyaw, sizeof(double) * 3);
    return RDET_OK;
}

int rdet2d_handle_odometry(rdet2d_t *h, double t, const double pos_xy[2], const double quat_zw[2],
                           double vx, double vy, double wz)
{
    if (!h || !pos_xy || !quat_zw) return RDET_ERR_INVALID;
    h->odom.push_back(Odom{t, pos_xy[0], pos_xy[1], 2 * std::atan2(quat_zw[0], quat_zw[1]), vx, vy, wz});   // pose_extrapolator.cc:28-32, yaw :36
    return RDET_OK;
}

int rdet2d_handle_scan(rdet2d_t *h, double stamp, float angle_min, float angle_max, float angle_increment,
                       float scan_time, float range_min, float range_max, const float *ranges,
                       const float *intensities, int N, float *centers_xy, int max_centers, int *K,
                       double *obs_time)
{
    if (!h || !K || N < 0 || (N > 0 && (!ranges || !intensities)) || max_centers < 0 || (max_centers > 0 && !centers_xy))
        return RDET_ERR_INVALID;
    *K = 0;
    if (obs_time) *obs_time = stamp;                                           // :26 (USE_CORRECT_TIME undefined)
    if (range_min < 0 || range_max <= range_min) return RDET_ERR_BAD_SCAN;      // :27-32
    if (angle_increment < 0.f && angle_max <= angle_min) return RDET_ERR_BAD_SCAN;   // :33-38
    DET_TRY(h, hipSetDevice(h->device));
    if (__atomic_load_n(&h->h_out->fin, __ATOMIC_ACQUIRE) != h->seq)   // the previous scan's kernel may still be compacting its point cloud
        DET_TRY(h, hipStreamSynchronize(h->stream));
    h->last_n_returns = 0;
    if (N == 0) return RDET_OK;
    if (N > h->max_beams || N > RDET2D_MAX_BEAMS) return RDET_ERR_CAPACITY;   // k_det2d: workgroup 0 holds the scan in LDS

    Det2dArgs A;
    std::memset(&A, 0, sizeof(A));
    A.intensity_min = h->opt.intensity_min;
    A.min_length = h->opt.reflector_min_length;
    A.length_error = h->opt.reflector_length_error;
    A.opt_range_min = h->opt.range_min; A.opt_range_max = h->opt.range_max;
    A.msg_range_min = range_min; A.msg_range_max = range_max;
    A.angle_increment = angle_increment;
    const double last_point_time = stamp;                                       // :48
    A.point_delta_t = (double)(scan_time / (float)N);                           // :49 (float / size_t)
    A.first_point_time = last_point_time - scan_time;                           // :50
    A.N = N;
    A.seq = ++h->seq;
    h->done_total += (unsigned)((N + RDET2D_GROUP - 1) / RDET2D_GROUP);
    A.done_target = h->done_total;
    A.is_circle = ((angle_max - angle_min - 2 * M_PI) < 1e-6) ? 1 : 0;          // :55
    A.max_centers = max_centers < RDET_MAX_CENTERS ? max_centers : RDET_MAX_CENTERS;
    A.s2b_x = (float)h->s2b[0]; A.s2b_y = (float)h->s2b[1]; A.s2b_a = (float)h->s2b[2];   // :54
    A.s2b_c = cosf(A.s2b_a); A.s2b_s = sinf(A.s2b_a);

    // TrimDataByTime(first_point_time) (:52-53 -> pose_extrapolator.cc:12-26)
    {
        size_t drop = 0;
        while (h->odom.size() - drop > 1 && h->odom[drop].time < A.first_point_time) ++drop;
        if (drop) h->odom.erase(h->odom.begin(), h->odom.begin() + (long)drop);
    }
    A.n_odom = (int)(h->odom.size() > 2 ? 2 : h->odom.size());
    if (!h->odom.empty()) { A.front = h->odom.front(); A.back = h->odom.back(); }

#ifdef RDET_DEBUG_MARKS
    const auto dbg_t0 = std::chrono::steady_clock::now();
    auto dbg_ns = [&]() { return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - dbg_t0).count(); };
#endif
    // the scan goes into pinned memory the kernel reads in place; the centres come back the same way: one launch and
    // one stream wait per scan, no copy engine
    std::memcpy(h->h_scan, ranges, sizeof(float) * (size_t)N);
    std::memcpy(h->h_scan + h->max_beams, intensities, sizeof(float) * (size_t)N);
    __atomic_thread_fence(__ATOMIC_SEQ_CST);              // write-combined stores drained before the doorbell
#ifdef RDET_DEBUG_MARKS
    h->h_marks[24] = dbg_ns();
#endif
    if (h->tab_N != N || h->tab_angle_min != angle_min || h->tab_inc != angle_increment) {
        // beam-angle table: the float32 accumulation of :51/:175 and its cos/sin (:68), host libm
        float *ta = h->h_table, *tc = ta + h->max_beams, *ts = tc + h->max_beams;
        float angle = angle_min;
        for (int i = 0; i < N; ++i) {
            ta[i] = angle; tc[i] = cosf(angle); ts[i] = sinf(angle);
            angle += angle_increment;
        }
        DET_TRY(h, hipMemcpyAsync(h->d_ang, ta, sizeof(float) * (size_t)N, hipMemcpyHostToDevice, h->stream));
        DET_TRY(h, hipMemcpyAsync(h->d_cos, tc, sizeof(float) * (size_t)N, hipMemcpyHostToDevice, h->stream));
        DET_TRY(h, hipMemcpyAsync(h->d_sin, ts, sizeof(float) * (size_t)N, hipMemcpyHostToDevice, h->stream));
        h->tab_N = N; h->tab_angle_min = angle_min; h->tab_inc = angle_increment;
    }
    Det2dBufs B;
    B.ranges = h->dv_scan; B.intens = h->dv_scan + h->max_beams;
    B.ang = h->d_ang; B.cosv = h->d_cos; B.sinv = h->d_sin;
    B.contrib = h->d_contrib; B.cmask = h->d_cmask; B.returns_all = h->d_returns_all; B.returns = h->d_returns;
    B.out = h->dv_out; B.done = h->d_done;
#ifdef RDET_DEBUG_MARKS
    B.marks = h->h_marks;
#endif
    hipLaunchKernelGGL(k_det2d, dim3(1 + (N + RDET2D_GROUP - 1) / RDET2D_GROUP), dim3(1024), 0, h->stream, A, B);
    DET_TRY(h, hipGetLastError());
#ifdef RDET_DEBUG_MARKS
    h->h_marks[25] = dbg_ns();
#endif
    // the kernel writes the centres and then this scan's number into host memory: poll for it instead of waiting for
    // the completion signal (the kernel's tail -- the point cloud for GetRangeData -- overlaps the caller)
    {
        const int *seq_word = &h->h_out->head.seq;
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (__atomic_load_n(seq_word, __ATOMIC_ACQUIRE) != A.seq) {
            if ((++spins & 0xfffffu) == 0) {                                      // every few hundred microseconds
                if (hipStreamQuery(h->stream) != hipErrorNotReady) {              // finished (or failed) without publishing?
                    DET_TRY(h, hipStreamSynchronize(h->stream));
                    if (__atomic_load_n(seq_word, __ATOMIC_ACQUIRE) == A.seq) break;
                    h->hip_error = "k_det2d finished without publishing its result";
                    return RDET_ERR_HIP;
                }
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
                    h->hip_error = "k_det2d: no result after 5 s";
                    return RDET_ERR_HIP;
                }
            }
        }
    }
#ifdef RDET_DEBUG_MARKS
    h->h_marks[26] = dbg_ns();
#endif
    const Det2dHead head = h->h_out->head;
    h->last_n_returns = head.n_returns;
    const int err = -(head.runs_err >> 16);
    if (err == RDET_ERR_HIP) {         // the kernel gave up waiting for its per-beam workgroups: drain, and start the counter afresh
        (void)hipStreamSynchronize(h->stream);
        (void)hipMemset(h->d_done, 0, sizeof(int));
        h->done_total = 0;
    }
    if (err) return err;
    *K = head.K;
    for (int c = 0; c < head.K; ++c) {
        const Det2dSlot *slot = &h->h_out->centers[c];
        unsigned spins = 0;
        while (__atomic_load_n(&slot->seq, __ATOMIC_ACQUIRE) != A.seq)
            if (++spins > 200000000u) { h->hip_error = "k_det2d: a centre never arrived"; return RDET_ERR_HIP; }
        centers_xy[2 * c] = slot->x; centers_xy[2 * c + 1] = slot->y;
    }
#ifdef RDET_DEBUG_MARKS
    h->h_marks[27] = (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - dbg_t0).count();
#endif
    return RDET_OK;
}

int rdet2d_get_range_data(rdet2d_t *h, float origin_xy[2], float *returns_xy, int cap_points, int *n_returns)
{
    if (!h || !n_returns) return RDET_ERR_INVALID;
    if (origin_xy) { origin_xy[0] = (float)h->s2b[0]; origin_xy[1] = (float)h->s2b[1]; }   // :243
    *n_returns = h->last_n_returns;
    if (returns_xy) {
        if (cap_points < h->last_n_returns) return RDET_ERR_BUFFER;
        if (h->last_n_returns > 0) {
            DET_TRY(h, hipSetDevice(h->device));
            DET_TRY(h, hipStreamSynchronize(h->stream));
            DET_TRY(h, hipMemcpy(returns_xy, h->d_returns, sizeof(float) * 2 * (size_t)h->last_n_returns,
                                 hipMemcpyDeviceToHost));
        }
    }
    return RDET_OK;
}

#ifdef RDET_DEBUG_MARKS
int rdet2d_debug_marks(rdet2d_t *h, unsigned long long *out32)
{
    std::memcpy(out32, h->h_marks, 8 * 32);
    return 0;
}
#endif

}  // extern "C"
