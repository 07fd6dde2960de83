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
// so one 1024-thread workgroup resolves a whole scan with three block-wide scans
// (last valid point, previous bright beam, run index), gates all closed runs in parallel
// (:147-156), resolves the first/last-run wrap logic on one lane (:178-236) and then
// de-skews every point (:239-270) and averages every accepted cluster in beam order with
// float32 running sums (:277-306), one lane per cluster.
//
// cos/sin of the float32-accumulated beam angle (Q15, :51,:175) come from a table the host
// fills with the same libm the reference would call, so run membership and gating see the
// same float32 points as a CPU build.  Everything is float32 where the reference is
// (geometry, point time stored in a Vector3f) and FP64 where it is (poses, odometry).
#include "../../include/rdet.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {

struct Odom {
    double time, px, py, qz, qw, vx, vy, wz;
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
    // sensor_to_base_link as Rigid2f + host-evaluated cos/sin of its angle
    float s2b_x, s2b_y, s2b_a, s2b_c, s2b_s;
    // pose extrapolator state: 0, 1 or 2 samples (front, back)
    int n_odom;
    Odom front, back;
};

struct Det2dOut {
    int K, n_returns, n_runs, err;
    float centers[2 * RDET_MAX_CENTERS];
};
// hand-over from the single-workgroup state machine (k_det2d) to the per-beam and per-cluster launches
struct Det2dMid {
    int K, off, n_cloud, pad;
    int seg[4 * RDET_MAX_CENTERS + 8];       // cluster segments: first0, last0, first1, last1
    double inv[5];                           // inverse scan-end pose (x, y, angle), cos / sin of that angle
};

struct Det2dBufs {
    const float *ranges, *intens, *ang, *cosv, *sinv;
    float2 *pt;            // per beam: point in base_link (valid beams)
    float2 *contrib;       // per beam of an accepted cluster: its de-skewed point in the scan-end frame (flags bit 8)
    float *pt_t;           // per beam: float32 point time
    int *lastvalid;        // last valid beam <= i  (point_cloud.back() at beam i), -1 if none
    int *cloud_idx;        // index of beam i in point_cloud (valid beams)
    int *prevb;            // previous bright beam (< i), -1 if none
    int *runid;            // run index of bright beam i
    unsigned char *flags;  // 1 valid, 2 bright (after the "a point exists" guard)
    int *run_first, *run_last, *run_acc;
    float2 *returns;
    Det2dOut *out;
    Det2dMid *mid;
};

// ---- Rigid2 algebra (rigid_transform.h:46-51,62-67,87-102), no FMA contraction -------------
__device__ static R2d r2_inverse(R2d r)
{
#pragma clang fp contract(off)
    R2d o;
    const double c = cos(-r.a), s = sin(-r.a);
    o.a = -r.a;
    o.x = -(c * r.x + (-s) * r.y);
    o.y = -(s * r.x + c * r.y);
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
    return r2f_apply_cs(cosf(r.a), sinf(r.a), r.x, r.y, px, py);
}

// ---- PoseExtrapolator (pose_extrapolator.cc:34-84,102-129) ------------------------------------
__device__ static R2d interpolator(const Odom &st, double time)
{
#pragma clang fp contract(off)
    R2d o;
    const double odom_yaw = 2 * atan2(st.qz, st.qw);
    if (st.time <= time) {
        const double delta_t = st.time - time;
        const double now_yaw = odom_yaw - st.wz * delta_t;
        const double c = cos(now_yaw), s = sin(now_yaw);
        o.x = st.px - st.vx * delta_t * c + st.vy * delta_t * s;
        o.y = st.py - st.vx * delta_t * s - st.vy * delta_t * c;
        o.a = now_yaw;
    } else {
        const double delta_t = time - st.time;
        const double now_yaw = odom_yaw - st.wz * delta_t;      // sign as in the reference (Q14)
        const double c = cos(now_yaw), s = sin(now_yaw);
        o.x = st.px + st.vx * delta_t * c - st.vy * delta_t * s;
        o.y = st.py + st.vx * delta_t * s + st.vy * delta_t * c;
        o.a = now_yaw;
    }
    return o;
}
__device__ static R2d extrapolator_pose(const Det2dArgs &A, double time)
{
    if (A.n_odom == 0) { R2d id = {0, 0, 0}; return id; }
    if (time <= A.front.time) return interpolator(A.front, time);
    return interpolator(A.back, time);    // t >= back, or in between: always the LAST sample (:76-82, Q14)
}

// ---- block-wide exclusive scans over 1024 per-thread values -----------------------------------
// in-wave scan by shuffles (no barrier), the 16 wave totals through LDS: two barriers per scan instead of
// the twenty of a Hillis-Steele scan over 1024 LDS slots
__device__ static int block_excl_sum(int v, int *lds, int *total)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int c = lds[w]; if (w < wave) base += c; tot += c; }
    if (total) *total = tot;
    __syncthreads();
    return base + incl - v;
}
__device__ static int block_excl_max(int v, int *lds, int *total)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl = max(incl, t);
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    int base = -1, tot = -1;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int c = lds[w]; if (w < wave) base = max(base, c); tot = max(tot, c); }
    if (total) *total = tot;
    __syncthreads();
    const int prev = __shfl_up(incl, 1, 64);                    // inclusive value of the previous lane
    return (lane == 0) ? base : max(base, prev);
}

__device__ static float gap_time_and_point(const Det2dArgs &A, const Det2dBufs &B, int j, float2 &p)
{
#pragma clang fp contract(off)
    // gap beam j (:115-130): re-projected from the NEXT bright beam's accumulated angle
    int inext = j + 1;
    while (inext < A.N && !(B.flags[inext] & 2)) ++inext;
    const float angle_gap = B.ang[inext] - A.angle_increment * (float)(inext - j);
    const float rg = B.ranges[j];
    p = r2f_apply_cs(A.s2b_c, A.s2b_s, A.s2b_x, A.s2b_y, rg * cosf(angle_gap), rg * sinf(angle_gap));
    return (float)(A.first_point_time + j * A.point_delta_t);
}

__global__ __launch_bounds__(1024) void k_det2d(Det2dArgs A, Det2dBufs B)
{
    __shared__ int lds[1024];
    __shared__ int s_tot[4];
    __shared__ int s_cl[4 * RDET_MAX_CENTERS + 8];   // cluster segments: first0,last0,first1,last1
    __shared__ double s_pose[8];                     // max_time_pose (x,y,a), cos/sin of its inverse angle
    const int tid = threadIdx.x;
    const int N = A.N;
    const int CH = (N + 1023) / 1024;
    const int b0 = tid * CH, b1 = min(N, b0 + CH);

    // ---- pass 1: points, validity, brightness (:63-83).  A thread owns CH <= 8 consecutive beams through
    // passes 1-3; their inputs and flags stay in registers (all loads of the pass in flight at once).
    constexpr int MAXCH = 8;
    float rg[MAXCH], cv[MAXCH], sv[MAXCH], it[MAXCH];
    unsigned char fl[MAXCH];
    int prevb_r[MAXCH];
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) {
        const int i = b0 + q;
        const bool in = q < CH && i < b1;
        rg[q] = in ? B.ranges[i] : 0.f; cv[q] = in ? B.cosv[i] : 0.f; sv[q] = in ? B.sinv[i] : 0.f; it[q] = in ? B.intens[i] : 0.f;
    }
    int cnt_valid = 0, last_valid = -1;
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) {
#pragma clang fp contract(off)
        const int i = b0 + q;
        fl[q] = 0;
        if (q >= CH || i >= b1) continue;
        const float range = rg[q];
        unsigned char f = 0;
        if (range >= A.msg_range_min && range <= A.msg_range_max) {
            f |= 1;
            const float nx = range * cv[q], ny = range * sv[q];
            B.pt[i] = r2f_apply_cs(A.s2b_c, A.s2b_s, A.s2b_x, A.s2b_y, nx, ny);
            B.pt_t[i] = (float)(A.first_point_time + i * A.point_delta_t);
            ++cnt_valid; last_valid = i;
        }
        if (A.opt_range_min <= range && range <= A.opt_range_max && (double)it[q] > A.intensity_min) f |= 4;
        fl[q] = f;
    }
    int n_cloud;
    const int cloud_base = block_excl_sum(cnt_valid, lds, &n_cloud);
    int lv = block_excl_max(last_valid, lds, nullptr);
    // ---- pass 2: point_cloud index / back(), guarded bright flag, previous bright beam
    int c = cloud_base, last_bright = -1;
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) {
        const int i = b0 + q;
        if (q >= CH || i >= b1) continue;
        unsigned char f = fl[q];
        if (f & 1) { B.cloud_idx[i] = c++; lv = i; }
        B.lastvalid[i] = lv;
        if ((f & 4) && lv >= 0) { f |= 2; last_bright = i; }
        fl[q] = f;
        B.flags[i] = f;
    }
    __syncthreads();
    int last_bright_all;
    int pb = block_excl_max(last_bright, lds, &last_bright_all);
    // ---- pass 3: run starts (:85-169) and run index
    int n_start = 0;
    unsigned start_mask = 0;
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) {
        const int i = b0 + q;
        prevb_r[q] = -1;
        if (q >= CH || i >= b1 || !(fl[q] & 2)) continue;
        B.prevb[i] = pb;
        prevb_r[q] = pb;
        bool start = pb < 0;
        if (!start && i - pb != 1) {
            const int nx = (i + 1 < N) ? i + 1 : i;
            const bool gap = (i - pb < 4) && (fabs((double)(rg[q] - B.ranges[pb])) < 0.3) &&
                             ((double)B.intens[nx] > A.intensity_min);                 // :111
            start = !gap;
        }
        if (start) start_mask |= 1u << q;
        n_start += start ? 1 : 0;
        pb = i;
    }
    int n_runs;
    int rid = block_excl_sum(n_start, lds, &n_runs);
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) {
        const int i = b0 + q;
        if (q >= CH || i >= b1 || !(fl[q] & 2)) continue;
        if (start_mask & (1u << q)) {      // start of run `rid`
            B.run_first[rid] = i;
            if (prevb_r[q] >= 0) B.run_last[rid - 1] = prevb_r[q];
            ++rid;
        }
        B.runid[i] = rid - 1;
    }
    if (tid == 0 && n_runs > 0) B.run_last[n_runs - 1] = last_bright_all;
    __syncthreads();

    // ---- gate the closed runs (:147-156), compact the accepted ones
    const int n_closed = (n_runs > 0) ? n_runs - 1 : 0;
    const int RCH = (n_closed + 1023) / 1024;
    const int r0 = tid * RCH, r1 = min(n_closed, r0 + RCH);
    int n_acc_local = 0;
    for (int r = r0; r < r1; ++r) {
#pragma clang fp contract(off)
        const int fi = B.run_first[r], li = B.run_last[r];
        const float2 pf = B.pt[B.lastvalid[fi]], pl = B.pt[B.lastvalid[li]];
        const float len = hypotf(pf.x - pl.x, pf.y - pl.y);
        const bool ok = (A.is_circle && fi == 0) || (fabs((double)len - A.min_length) < A.length_error);
        B.run_acc[r] = ok ? 1 : 0;
        n_acc_local += ok ? 1 : 0;
    }
    int n_acc;
    int cidx = block_excl_sum(n_acc_local, lds, &n_acc);
    for (int r = r0; r < r1; ++r) {
        if (!B.run_acc[r]) continue;
        if (cidx < RDET_MAX_CENTERS) {
            s_cl[4 * cidx + 0] = B.run_first[r]; s_cl[4 * cidx + 1] = B.run_last[r];
            s_cl[4 * cidx + 2] = -1; s_cl[4 * cidx + 3] = -1;
        }
        ++cidx;
    }
    __syncthreads();

    // ---- last / first reflector (:178-236) and the scan-end pose, one lane
    if (tid == 0) {
#pragma clang fp contract(off)
        int n_cl = n_acc, off = 0, err = 0;
        if (n_acc > RDET_MAX_CENTERS) { err = RDET_ERR_CAPACITY; n_cl = RDET_MAX_CENTERS; }
        if (n_runs > 0 && !err) {
            const int Lr = n_runs - 1;
            const int lf = B.run_first[Lr], ll = B.run_last[Lr];
            const float2 last_first_pt = B.pt[B.lastvalid[lf]], last_pt = B.pt[B.lastvalid[ll]];
            const float len = hypotf(last_first_pt.x - last_pt.x, last_first_pt.y - last_pt.y);
            const bool len_ok = fabs((double)len - A.min_length) < A.length_error;
            if (n_cl > 0) {
                const int first_id = s_cl[0];
                const float2 first_pt = B.pt[B.lastvalid[s_cl[0]]];
                const float2 first_last_pt = B.pt[B.lastvalid[s_cl[1]]];
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
        s_tot[0] = K; s_tot[1] = off; s_tot[2] = err;
        // scan-end pose (:252-253, :299)
        if (n_cloud > 0) {
            int last_valid_all = N - 1;
            while (last_valid_all >= 0 && !(B.flags[last_valid_all] & 1)) --last_valid_all;
            const R2d mtp = extrapolator_pose(A, (double)B.pt_t[last_valid_all]);
            const R2d inv = r2_inverse(mtp);
            s_pose[0] = inv.x; s_pose[1] = inv.y; s_pose[2] = inv.a;
            s_pose[3] = cos(inv.a); s_pose[4] = sin(inv.a);
        }
        B.out->K = K; B.out->n_returns = n_cloud; B.out->n_runs = n_runs; B.out->err = err;
    }
    __syncthreads();
    // hand-over: everything after this point is per beam or per cluster and runs as two wide launches (in this one
    // workgroup it was FP64 trigonometry for 3600 beams on a single CU: 55 of the kernel's 95 k cycles)
    Det2dMid *M = B.mid;
    if (tid == 0) { M->K = s_tot[0]; M->off = s_tot[1]; M->n_cloud = n_cloud; }
    if (tid < 5) M->inv[tid] = s_pose[tid];
    for (int q = tid; q < 4 * RDET_MAX_CENTERS + 8; q += 1024) M->seg[q] = s_cl[q];
}

// ---- per beam: de-skew into the scan-end frame (:246-258) and, for the beams of an accepted cluster, their
// contribution to the centre (:277-299).  A beam finds its cluster by bisection: the clusters' first segments are in
// ascending beam order (run order); the wrapped second segment of the first cluster is tested on its own.
__global__ __launch_bounds__(256) void k_det2d_beams(Det2dArgs A, Det2dBufs B)
{
    __shared__ int s_cl[4 * RDET_MAX_CENTERS + 8];
    const Det2dMid *M = B.mid;
    const int n_cloud = M->n_cloud;
    if (n_cloud == 0) return;
    const int K = M->K, off = M->off, N = A.N;
    for (int q = threadIdx.x; q < 4 * RDET_MAX_CENTERS + 8; q += 256) s_cl[q] = M->seg[q];
    const R2d inv = {M->inv[0], M->inv[1], M->inv[2]};
    const double inv_c = M->inv[3], inv_s = M->inv[4];
    const R2f to_base = r2_cast(inv);
    const float tb_c = cosf(to_base.a), tb_s = sinf(to_base.a);
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int jc = min(j, N - 1);
    const unsigned char f = B.flags[jc];
    const int lvj = B.lastvalid[jc], ci = B.cloud_idx[jc];
    const float rgj = B.ranges[jc], tj = B.pt_t[jc];
    const float2 pj = B.pt[jc];
    __syncthreads();
    if (j >= N) return;
    if (f & 1) {                                                              // de-skew (:246-258)
        const R2d pose = extrapolator_pose(A, (double)tj);
        const R2f rel = r2_cast(r2_mul_cs(inv_c, inv_s, inv, pose));
        B.returns[ci] = r2f_apply(rel, pj.x, pj.y);
    }
    if (K <= 0) return;
    const int k0 = off, k1 = off + K;
    bool mem = s_cl[4 * k0 + 2] >= 0 && j >= s_cl[4 * k0 + 2] && j <= s_cl[4 * k0 + 3];
    if (!mem) {
        int lo = k0, hi = k1 - 1;                                             // last cluster whose first beam is <= j
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_cl[4 * mid] <= j) lo = mid; else hi = mid - 1; }
        mem = s_cl[4 * lo] <= j && j <= s_cl[4 * lo + 1];
    }
    if (!mem) return;
    {
#pragma clang fp contract(off)
        float2 p; float t;
        if (f & 2) { p = B.pt[lvj]; t = B.pt_t[lvj]; }
        else if (isinf(rgj)) return;                                          // :120-121
        else t = gap_time_and_point(A, B, j, p);
        const R2f pose = r2_cast(extrapolator_pose(A, (double)t));             // :287,:293
        const float2 po = r2f_apply(pose, p.x, p.y);
        B.contrib[j] = r2f_apply_cs(tb_c, tb_s, to_base.x, to_base.y, po.x, po.y);
        B.flags[j] = f | 8;                                                   // contributes (this thread is the beam's only writer)
    }
}

// ---- per cluster: the float32 running sum in beam order that the reference takes (:300-305), one wave per cluster
__global__ __launch_bounds__(256) void k_det2d_sums(Det2dArgs A, Det2dBufs B)
{
#pragma clang fp contract(off)
    const Det2dMid *M = B.mid;
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (M->n_cloud == 0 || c >= M->K) return;
    const int k = c + M->off, N = A.N;
    float cx = 0.f, cy = 0.f;
    int count = 0;
    for (int seg = 0; seg < 2; ++seg) {
        const int fi = M->seg[4 * k + 2 * seg], li = M->seg[4 * k + 2 * seg + 1];
        if (fi < 0) continue;
        for (int j0 = fi; j0 <= li; j0 += 64) {
            const int j = j0 + lane;
            const float2 cv2 = B.contrib[min(j, N - 1)];                          // both loads unconditional: one round trip
            const bool mem = j <= li && (B.flags[min(j, N - 1)] & 8);
            const float2 v = mem ? cv2 : make_float2(0.f, 0.f);
            unsigned long long mask = __ballot(mem);
            count += __popcll(mask);
            while (mask) {
                const int b = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                cx += __shfl(v.x, b, 64);
                cy += __shfl(v.y, b, 64);
            }
        }
    }
    if (lane == 0) {
        B.out->centers[2 * c] = cx / (float)count;                               // :305
        B.out->centers[2 * c + 1] = cy / (float)count;
    }
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
    float *d_ranges, *d_intens, *d_ang, *d_cos, *d_sin, *d_pt_t;
    float2 *d_pt, *d_returns, *d_contrib;
    int *d_lastvalid, *d_cloud_idx, *d_prevb, *d_runid, *d_run_first, *d_run_last, *d_run_acc;
    unsigned char *d_flags;
    Det2dOut *d_out, *h_out;           // h_out pinned
    Det2dMid *d_mid;
    float *h_stage;                    // pinned: ranges | intensities | ang | cos | sin
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
        DET_TRY(h, hipMalloc(&h->d_ranges, 4 * nb)); DET_TRY(h, hipMalloc(&h->d_intens, 4 * nb));
        DET_TRY(h, hipMalloc(&h->d_ang, 4 * nb)); DET_TRY(h, hipMalloc(&h->d_cos, 4 * nb));
        DET_TRY(h, hipMalloc(&h->d_sin, 4 * nb)); DET_TRY(h, hipMalloc(&h->d_pt_t, 4 * nb));
        DET_TRY(h, hipMalloc(&h->d_pt, 8 * nb)); DET_TRY(h, hipMalloc(&h->d_returns, 8 * nb)); DET_TRY(h, hipMalloc(&h->d_contrib, 8 * nb));
        DET_TRY(h, hipMalloc(&h->d_lastvalid, 4 * nb)); DET_TRY(h, hipMalloc(&h->d_cloud_idx, 4 * nb));
        DET_TRY(h, hipMalloc(&h->d_prevb, 4 * nb)); DET_TRY(h, hipMalloc(&h->d_runid, 4 * nb));
        DET_TRY(h, hipMalloc(&h->d_run_first, 4 * nb)); DET_TRY(h, hipMalloc(&h->d_run_last, 4 * nb));
        DET_TRY(h, hipMalloc(&h->d_run_acc, 4 * nb)); DET_TRY(h, hipMalloc(&h->d_flags, nb));
        DET_TRY(h, hipMalloc(&h->d_out, sizeof(Det2dOut)));
        DET_TRY(h, hipMalloc(&h->d_mid, sizeof(Det2dMid)));
        DET_TRY(h, hipHostMalloc(&h->h_out, sizeof(Det2dOut)));
        DET_TRY(h, hipHostMalloc(&h->h_stage, 4 * nb * 5));
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
    void *ptrs[] = {h->d_ranges, h->d_intens, h->d_ang, h->d_cos, h->d_sin, h->d_pt_t, h->d_pt, h->d_returns, h->d_contrib,
                    h->d_lastvalid, h->d_cloud_idx, h->d_prevb, h->d_runid, h->d_run_first, h->d_run_last,
                    h->d_run_acc, h->d_flags, h->d_out, h->d_mid};
    for (void *p : ptrs) (void)hipFree(p);
    if (h->h_out) (void)hipHostFree(h->h_out);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
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
    h->odom.push_back(Odom{t, pos_xy[0], pos_xy[1], quat_zw[0], quat_zw[1], vx, vy, wz});   // pose_extrapolator.cc:28-32
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
    h->last_n_returns = 0;
    if (N == 0) return RDET_OK;
    if (N > h->max_beams || N > 8192) return RDET_ERR_CAPACITY;      // k_det2d: one workgroup, <= 8 beams per thread
    DET_TRY(h, hipSetDevice(h->device));

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

    float *st_r = h->h_stage, *st_i = st_r + h->max_beams;
    std::memcpy(st_r, ranges, sizeof(float) * (size_t)N);
    std::memcpy(st_i, intensities, sizeof(float) * (size_t)N);
    DET_TRY(h, hipMemcpyAsync(h->d_ranges, st_r, sizeof(float) * (size_t)N, hipMemcpyHostToDevice, h->stream));
    DET_TRY(h, hipMemcpyAsync(h->d_intens, st_i, sizeof(float) * (size_t)N, hipMemcpyHostToDevice, h->stream));
    if (h->tab_N != N || h->tab_angle_min != angle_min || h->tab_inc != angle_increment) {
        // beam-angle table: the float32 accumulation of :51/:175 and its cos/sin (:68), host libm
        float *ta = st_i + h->max_beams, *tc = ta + h->max_beams, *ts = tc + h->max_beams;
        float angle = angle_min;
        for (int i = 0; i < N; ++i) {
            ta[i] = angle; tc[i] = cosf(angle); ts[i] = sinf(angle);
            angle += angle_increment;
        }
        DET_TRY(h, hipMemcpyAsync(h->d_ang, ta, sizeof(float) * (size_t)N, hipMemcpyHostToDevice, h->stream));
        DET_TRY(h, hipMemcpyAsync(h->d_cos, tc, sizeof(float) * (size_t)N, hipMemcpyHostToDevice, h->stream));
        DET_TRY(h, hipMemcpyAsync(h->d_sin, ts, sizeof(float) * (size_t)N, hipMemcpyHostToDevice, h->stream));
        DET_TRY(h, hipStreamSynchronize(h->stream));     // the staging area is reused by the next scan
        h->tab_N = N; h->tab_angle_min = angle_min; h->tab_inc = angle_increment;
    }
    Det2dBufs B;
    B.ranges = h->d_ranges; B.intens = h->d_intens; B.ang = h->d_ang; B.cosv = h->d_cos; B.sinv = h->d_sin;
    B.pt = h->d_pt; B.pt_t = h->d_pt_t; B.lastvalid = h->d_lastvalid; B.cloud_idx = h->d_cloud_idx;
    B.prevb = h->d_prevb; B.runid = h->d_runid; B.flags = h->d_flags;
    B.run_first = h->d_run_first; B.run_last = h->d_run_last; B.run_acc = h->d_run_acc;
    B.returns = h->d_returns; B.contrib = h->d_contrib; B.out = h->d_out; B.mid = h->d_mid;
    hipLaunchKernelGGL(k_det2d, dim3(1), dim3(1024), 0, h->stream, A, B);
    hipLaunchKernelGGL(k_det2d_beams, dim3((N + 255) / 256), dim3(256), 0, h->stream, A, B);
    hipLaunchKernelGGL(k_det2d_sums, dim3(RDET_MAX_CENTERS / 4), dim3(256), 0, h->stream, A, B);
    DET_TRY(h, hipMemcpyAsync(h->h_out, h->d_out, sizeof(Det2dOut), hipMemcpyDeviceToHost, h->stream));
    DET_TRY(h, hipStreamSynchronize(h->stream));
    h->last_n_returns = h->h_out->n_returns;
    if (h->h_out->err) return h->h_out->err;
    *K = h->h_out->K;
    if (*K > 0) std::memcpy(centers_xy, h->h_out->centers, sizeof(float) * 2 * (size_t)*K);
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
            DET_TRY(h, hipMemcpy(returns_xy, h->d_returns, sizeof(float) * 2 * (size_t)h->last_n_returns,
                                 hipMemcpyDeviceToHost));
        }
    }
    return RDET_OK;
}

}  // extern "C"
