// ekf_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the EKF hot path.
//
// One HandleObservationMessage (reference reflector_ekf_slam.cc:229-368) is the
// kernel chain   front -> gather -> solve -> gain -> downdate -> augment
// on the handle's stream, with every size (n, m, match lists) resident in HBM
// (RekfCtl) so the host never waits for the device between scans.
//
//   k_front     predict (cc:154-206) + ReflectorMatch (cc:370-455) + H rows/z/zhat (cc:248-304)
//   k_gather    W = P H^T and (H P)^T using the <=5 structural non-zeros of each H row (cc:305,308)
//   k_solve     S = H W + Q, S^-1 by in-register Gauss-Jordan, y = S^-1 (z - zhat) (cc:305)
//   k_gain      Kn = -W S^-1 (FP64 MFMA), mu += W y, theta wrap (cc:306-307)
//   k_downdate  P += Kn (H P) : the FP64 MFMA, LDS-tiled rank-m downdate (cc:308) -- the roofline kernel
//   k_augment   new landmark means and covariance blocks (cc:311-364)
//
// Like the reference, P is never symmetrised: W = P H^T is gathered from the
// COLUMNS of P and HP^T from its ROWS.  (Taking H P := (P H^T)^T looks harmless
// but is unstable: with it the antisymmetric round-off part A of P evolves as
// A + (P G) A (G P) instead of the reference's contraction (I - P G) A (I - G P),
// G = H^T S^-1 H, and grows exponentially -- measured 1e-17 -> 1e-5 in 300 scans.)
//
// Deliberate, stated deviation from the literal Eigen expressions (FP64 round-off
// level, far inside the 1e-5 m parity bar):
//   * S^-1 by Gauss-Jordan without pivoting (S = H P H^T + Q is SPD) instead of
//     Eigen's partial-pivot LU.
#include "ekf_dev.h"

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

#define WAVE 64

// ----------------------------------------------------------------------------
// small scalar pieces: kept free of FMA contraction so that the float32
// roundings the reference performs (cc:389-393, :431-433, :327-331) see the
// same doubles as a plain x86-64 build of the reference.
// ----------------------------------------------------------------------------
struct Motion {
    double d[3];
    double a, b;
    double V[9];
};

__device__ static void motion_terms(const RekfFrontArgs &A, double theta, Motion &mo)
{
#pragma clang fp contract(off)
    const double vx = A.vt[0], vy = A.vt[1], w = A.vt[2], dt = A.dt;
    double Gu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double Qu[3];
    int q;
    if (A.model == 0) {                                   // DIFF  cc:156-183
        const double delta_theta = w * dt;
        const double half = theta + delta_theta / 2;
        const double ch = cos(half), sh = sin(half);
        mo.d[0] = vx * dt * ch;
        mo.d[1] = vx * dt * sh;
        mo.d[2] = delta_theta;
        mo.a = -vx * dt * sh;
        mo.b = vx * dt * ch;
        q = 2;
        Gu[0] = dt * ch; Gu[1] = -vx * dt * dt * sh / 2;
        Gu[3] = dt * sh; Gu[4] = vx * dt * dt * ch / 2;
        Gu[6] = 0;       Gu[7] = dt;
        Qu[0] = A.lin_cov; Qu[1] = A.ang_cov; Qu[2] = 0;
    } else {                                              // OMNI  cc:184-205
        const double delta_theta = w * dt;
        const double ct = cos(theta), st = sin(theta);
        mo.d[0] = vx * dt * ct - vy * dt * st;
        mo.d[1] = vx * dt * st + vy * dt * ct;
        mo.d[2] = delta_theta;
        mo.a = -vx * dt * st - vy * dt * ct;
        mo.b = vx * dt * ct - vy * dt * st;
        q = 3;
        Gu[0] = dt * ct; Gu[1] = -dt * st; Gu[2] = 0.;
        Gu[3] = dt * st; Gu[4] = dt * ct;  Gu[5] = 0.;
        Gu[6] = 0.;      Gu[7] = 0.;       Gu[8] = dt;
        Qu[0] = A.lin_cov; Qu[1] = A.lin_cov; Qu[2] = A.ang_cov;
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < q; ++k)
                s += Gu[i * 3 + k] * Qu[k] * Gu[j * 3 + k];
            mo.V[i * 3 + j] = s;
        }
}

// 3x3 pose block of G P G^T + Gu Qu Gu^T (row ops, then column ops, then + V),
// in place on a column-major 3x3 with leading dimension ld.
__device__ static void corner_predict(double *P, int ld, const Motion &mo)
{
#pragma clang fp contract(off)
    for (int c = 0; c < 3; ++c) {
        const double p2 = P[2 + (size_t)c * ld];
        P[0 + (size_t)c * ld] = P[0 + (size_t)c * ld] + mo.a * p2;
        P[1 + (size_t)c * ld] = P[1 + (size_t)c * ld] + mo.b * p2;
    }
    for (int r = 0; r < 3; ++r) {
        const double p2 = P[r + (size_t)2 * ld];
        P[r + (size_t)0 * ld] = P[r + (size_t)0 * ld] + mo.a * p2;
        P[r + (size_t)1 * ld] = P[r + (size_t)1 * ld] + mo.b * p2;
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            P[i + (size_t)j * ld] += mo.V[i * 3 + j];
}

__device__ static void obs_to_global(double x, double y, double th, float px, float py,
                                     float &gx, float &gy)
{
#pragma clang fp contract(off)
    // cc:389-393 / cc:327-331: evaluated in double, rounded to float32 on assignment
    gx = (float)((double)px * cos(th) - (double)py * sin(th) + x);
    gy = (float)((double)px * sin(th) + (double)py * cos(th) + y);
}

__device__ static double yaw_innovation(double delta_theta)
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

// wave-wide arg-min of (d, j) with the "first minimum in index order" rule
// (cc:414-419 sorts with '<=' and takes front(); ties are UB there).
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
// k_front: one workgroup of 16 waves.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_front(RekfDev d, RekfFrontArgs A)
{
    __shared__ Motion mo;
    __shared__ double pose[3];
    __shared__ int s_n;
    __shared__ int s_kind[REKF_MAX_OBS_DEV];   // 0 map match, 1 state match, 2 new
    __shared__ int s_idx[REKF_MAX_OBS_DEV];
    __shared__ int s_pair_obs[REKF_MAX_OBS_DEV], s_pair_id[REKF_MAX_OBS_DEV], s_pair_state[REKF_MAX_OBS_DEV];
    __shared__ int s_counts[4];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    RekfCtl *ctl = d.ctl;
    double *P = d.P;
    double *mu = d.mu;
    const size_t ld = (size_t)d.ld;

    if (tid == 0) {
        s_n = ctl->n;
        motion_terms(A, mu[2], mo);
    }
    __syncthreads();
    const int n = s_n;

    // ---- Predict, covariance: P <- G P G^T + Gu Qu Gu^T.  G = I + a e0 e2^T + b e1 e2^T
    // touches rows 0,1 and columns 0,1 only (the reference multiplies dense n x n, cc:178/202).
    {
#pragma clang fp contract(off)
        const double a = mo.a, b = mo.b;
        for (int idx = tid; idx < n; idx += 1024) {
            if (idx >= 3) {
                const double p2 = P[idx + 2 * ld];                 // column part (coalesced)
                P[idx + 0 * ld] = P[idx + 0 * ld] + a * p2;
                P[idx + 1 * ld] = P[idx + 1 * ld] + b * p2;
                const double q2 = P[2 + idx * ld];                 // row part (strided)
                P[0 + idx * ld] = P[0 + idx * ld] + a * q2;
                P[1 + idx * ld] = P[1 + idx * ld] + b * q2;
            }
        }
        if (tid == 0) {
            corner_predict(P, d.ld, mo);
            // mean (cc:180-181 / :204-205)
            const double x = mu[0] + mo.d[0], y = mu[1] + mo.d[1];
            double th = mu[2] + mo.d[2];
            th = atan2(sin(th), cos(th));
            mu[0] = x; mu[1] = y; mu[2] = th;
            pose[0] = x; pose[1] = y; pose[2] = th;
        }
    }
    __syncthreads();
    if (!A.is_obs) return;                    // odometry path: HandleOdometryMessage cc:208-223

    const int K = A.K;
    if (K <= 0) {                             // cc:235-236 (empty cloud): record cleared
        if (tid == 0) {
            ctl->K = 0; ctl->n_state = 0; ctl->n_map = 0; ctl->n_new = 0; ctl->m = 0; ctl->m_pad = 0;
        }
        return;
    }

    // ---- ReflectorMatch (cc:370-455): one wave per observation, lanes over candidates
    const int L = (n - 3) / 2;
    const int M_ = d.M_map;
    for (int i = wave; i < K; i += 16) {
        float gx, gy;
        obs_to_global(pose[0], pose[1], pose[2], A.obs[2 * i], A.obs[2 * i + 1], gx, gy);
        int kind = 2, best_j = -1;
        if (M_ > 0) {                                              // cc:401-425
#pragma clang fp contract(off)
            double best = 0; int bj = -1;
            for (int j = lane; j < M_; j += WAVE) {
                const double *S = d.map_cov + 4 * (size_t)j;
                const float ex = d.map_xy[2 * j] - gx;             // float32 subtract (cc:408)
                const float ey = d.map_xy[2 * j + 1] - gy;
                const double dx = (double)ex, dy = (double)ey;
                const double t0 = dx * S[0] + dy * S[2];
                const double t1 = dx * S[1] + dy * S[3];
                const double dist = sqrt(t0 * dx + t1 * dy);       // delta Sigma delta^T (cc:411)
                if (bj < 0 || dist < best) { best = dist; bj = j; }
            }
            wave_argmin(best, bj);
            if (bj >= 0 && best < 0.05) { kind = 0; best_j = bj; } // cc:420
        }
        if (kind == 2 && L > 0) {                                  // cc:426-451
#pragma clang fp contract(off)
            double best = 0; int bj = -1;
            for (int j = lane; j < L; j += WAVE) {
                const float lx = (float)mu[3 + 2 * j];             // cc:431
                const float ly = (float)mu[4 + 2 * j];
                const float ex = gx - lx;                          // cc:433
                const float ey = gy - ly;
                const double dx = (double)ex, dy = (double)ey;
                const double dist = sqrt(dx * dx + dy * dy);       // cc:437
                if (bj < 0 || dist < best) { best = dist; bj = j; }
            }
            wave_argmin(best, bj);
            if (bj >= 0 && best < 0.6) { kind = 1; best_j = bj; }  // cc:446
        }
        if (lane == 0) { s_kind[i] = kind; s_idx[i] = best_j; }
    }
    __syncthreads();

    // ---- ordered compaction into the three lists (obs order preserved)
    if (wave == 0) {
        const int kind = (lane < K) ? s_kind[lane] : -1;
        const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const unsigned long long ms = __ballot(kind == 1);
        const unsigned long long mm = __ballot(kind == 0);
        const unsigned long long mn = __ballot(kind == 2);
        const int M = __popcll(ms), Mm = __popcll(mm);
        int N2 = __popcll(mn);
        const int room = (d.n_max - n) / 2;
        if (N2 > room) {                                           // capacity guard (ours)
            if (lane == 0) atomicOr(&ctl->err, REKF_FLAG_CAPACITY);
            N2 = room;
        }
        if (kind == 1) {
            const int p = __popcll(ms & lt);
            ctl->state_pairs[2 * p] = lane; ctl->state_pairs[2 * p + 1] = s_idx[lane];
            s_pair_obs[p] = lane; s_pair_id[p] = s_idx[lane]; s_pair_state[p] = 1;
        } else if (kind == 0) {
            const int p = __popcll(mm & lt);
            ctl->map_pairs[2 * p] = lane; ctl->map_pairs[2 * p + 1] = s_idx[lane];
            s_pair_obs[M + p] = lane; s_pair_id[M + p] = s_idx[lane]; s_pair_state[M + p] = 0;
        } else if (kind == 2) {
            const int p = __popcll(mn & lt);
            if (p < N2) ctl->new_ids[p] = lane;
        }
        if (lane == 0) {
            const int MM = M + Mm;
            const int m = (MM > 0) ? 2 * MM + (A.has_gps ? 3 : 0) : 0;
            ctl->K = K; ctl->n_state = M; ctl->n_map = Mm; ctl->n_new = N2;
            ctl->m = m; ctl->m_pad = (m + 15) & ~15;
            s_counts[0] = M; s_counts[1] = Mm; s_counts[2] = m;
        }
    }
    __syncthreads();

    // ---- H rows, z - zhat, Q (cc:248-304): row pair p per thread
    const int M = s_counts[0], MM = s_counts[0] + s_counts[1];
    if (tid < MM) {
#pragma clang fp contract(off)
        const int p = tid;
        const int local_id = s_pair_obs[p], global_id = s_pair_id[p], is_state = s_pair_state[p];
        const double c = cos(pose[2]), s = sin(pose[2]);           // cc:252-253
        const double z0 = (double)A.obs[2 * local_id], z1 = (double)A.obs[2 * local_id + 1];
        double lx, ly;
        if (is_state) { lx = mu[3 + 2 * global_id]; ly = mu[4 + 2 * global_id]; }
        else { lx = (double)d.map_xy[2 * global_id]; ly = (double)d.map_xy[2 * global_id + 1]; }
        const double dx = lx - pose[0], dy = ly - pose[1];         // cc:267-268
        const double zh0 = dx * c + dy * s, zh1 = -dx * s + dy * c; // cc:269-270
        const int r0 = 2 * p, r1 = 2 * p + 1;
        ctl->ha[r0][0] = -c; ctl->ha[r0][1] = -s; ctl->ha[r0][2] = -dx * s + dy * c;   // A_i cc:272-273
        ctl->ha[r1][0] = s;  ctl->ha[r1][1] = -c; ctl->ha[r1][2] = -dx * c - dy * s;
        ctl->hb[r0][0] = c;  ctl->hb[r0][1] = s;                   // B cc:255 (state rows only, cc:275)
        ctl->hb[r1][0] = -s; ctl->hb[r1][1] = c;
        const int col = is_state ? 3 + 2 * global_id : -1;
        ctl->hcol[r0] = col; ctl->hcol[r1] = col;
        ctl->dz[r0] = z0 - zh0; ctl->dz[r1] = z1 - zh1;
        ctl->qd[r0] = A.obs_cov; ctl->qd[r1] = A.obs_cov;          // cc:276 / :302
    }
    if (tid == 0 && A.has_gps && MM > 0) {                         // gps.cc:305-332
#pragma clang fp contract(off)
        const int r0 = 2 * MM;
        for (int k = 0; k < 3; ++k) {
            ctl->ha[r0 + k][0] = (k == 0); ctl->ha[r0 + k][1] = (k == 1); ctl->ha[r0 + k][2] = (k == 2);
            ctl->hb[r0 + k][0] = 0; ctl->hb[r0 + k][1] = 0; ctl->hcol[r0 + k] = -1;
        }
        ctl->dz[r0] = A.gps[0] - pose[0];
        ctl->dz[r0 + 1] = A.gps[1] - pose[1];
        ctl->dz[r0 + 2] = yaw_innovation(A.gps[2] - pose[2]);
        ctl->qd[r0] = 0.05 * 0.05; ctl->qd[r0 + 1] = 0.05 * 0.05; ctl->qd[r0 + 2] = 0.017 * 0.017;
    }
    (void)M;
}

// ----------------------------------------------------------------------------
// k_gather: W(c, r) = sum_k P(c,k) H(r,k)  (columns of P: coalesced) and
// HPt(c, r) = (H P)(r, c) = sum_k H(r,k) P(k,c)  (rows of P: each thread walks
// its own column c); thread per state index c, blockIdx.y strides over row pairs.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather(RekfDev d)
{
    const RekfCtl *ctl = d.ctl;
    const int m = ctl->m;
    if (m == 0) return;
    const int n = ctl->n, m_pad = ctl->m_pad;
    const size_t ld = (size_t)d.ld;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d.ld) return;
    const bool valid = c < n;
    const double *P = d.P;
    double p0 = 0, p1 = 0, p2 = 0, q0 = 0, q1 = 0, q2 = 0;
    const double *Pc = P + (size_t)c * ld;                       // column c
    if (valid) {
        p0 = P[c]; p1 = P[c + ld]; p2 = P[c + 2 * ld];
        q0 = Pc[0]; q1 = Pc[1]; q2 = Pc[2];
    }
    for (int pr = blockIdx.y; pr < m_pad / 2; pr += gridDim.y) {
        for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * pr + rr;
            double v = 0, u = 0;
            if (valid && r < m) {
                const int col = ctl->hcol[r];
                const double h0 = ctl->ha[r][0], h1 = ctl->ha[r][1], h2 = ctl->ha[r][2];
                v = p0 * h0; v += p1 * h1; v += p2 * h2;
                u = h0 * q0; u += h1 * q1; u += h2 * q2;
                if (col >= 0) {
                    const double g0 = ctl->hb[r][0], g1 = ctl->hb[r][1];
                    v += P[c + (size_t)col * ld] * g0;
                    v += P[c + (size_t)(col + 1) * ld] * g1;
                    u += g0 * Pc[col];
                    u += g1 * Pc[col + 1];
                }
            }
            d.W[c + (size_t)r * ld] = v;
            d.HPt[c + (size_t)r * ld] = u;
        }
    }
}

// ----------------------------------------------------------------------------
// k_solve: S = H W + Q (m x m), S^-1 by Gauss-Jordan, y = S^-1 dz.
// 1024 threads as a 32x32 grid; thread (ti,tj) keeps S(ti+32a, tj+32b) in
// registers; per elimination step only row k and column k travel through LDS
// (ping-pong buffers, one barrier per step).
// ----------------------------------------------------------------------------
template <int NB>
__device__ static void solve_body(const RekfDev &d, int m, double (*rowbuf)[REKF_MR_PAD],
                                  double (*colbuf)[REKF_MR_PAD])
{
    RekfCtl *ctl = d.ctl;
    const int tid = threadIdx.x;
    const int ti = tid >> 5, tj = tid & 31;
    const size_t ld = (size_t)d.ld;
    double S[NB][NB];
#pragma unroll
    for (int a = 0; a < NB; ++a) {
        const int i = ti + 32 * a;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int j = tj + 32 * b;
            double v = (i == j) ? 1.0 : 0.0;
            if (i < m && j < m) {
                const double *w = d.W + (size_t)j * ld;          // column j of W
                const int col = ctl->hcol[i];
                v = ctl->ha[i][0] * w[0];
                v += ctl->ha[i][1] * w[1];
                v += ctl->ha[i][2] * w[2];
                if (col >= 0) {
                    v += ctl->hb[i][0] * w[col];
                    v += ctl->hb[i][1] * w[col + 1];
                }
                if (i == j) v += ctl->qd[i];
            }
            S[a][b] = v;
        }
    }
    bool bad = false;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        for (int kk = 0; kk < 32; ++kk) {
            const int k = 32 * kb + kk;
            if (k >= m) break;
            const int buf = k & 1;
            if (ti == kk) {
#pragma unroll
                for (int b = 0; b < NB; ++b) rowbuf[buf][tj + 32 * b] = S[kb][b];
            }
            if (tj == kk) {
#pragma unroll
                for (int a = 0; a < NB; ++a) colbuf[buf][ti + 32 * a] = S[a][kb];
            }
            __syncthreads();
            const double piv = rowbuf[buf][k];
            if (!(piv > 0.0)) bad = true;
            const double p = 1.0 / piv;
#pragma unroll
            for (int a = 0; a < NB; ++a) {
                const int i = ti + 32 * a;
                const double f = colbuf[buf][i] * p;
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const int j = tj + 32 * b;
                    const double rj = rowbuf[buf][j];
                    double v;
                    if (i == k) v = (j == k) ? p : rj * p;
                    else if (j == k) v = -f;
                    else v = S[a][b] - f * rj;
                    S[a][b] = v;
                }
            }
        }
    }
    if (bad && tid == 0) atomicOr(&ctl->err, REKF_FLAG_SINGULAR);
    // S^-1 out (pad rows/cols are the identity) and y = S^-1 dz
#pragma unroll
    for (int a = 0; a < NB; ++a) {
        const int i = ti + 32 * a;
        double acc = 0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int j = tj + 32 * b;
            d.Sinv[i + (size_t)j * REKF_MR_PAD] = S[a][b];
            if (j < m) acc += S[a][b] * ctl->dz[j];
        }
        for (int off = 16; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 32);
        if (tj == 0) d.y[i] = (i < m) ? acc : 0.0;
    }
}

__global__ __launch_bounds__(1024) void k_solve(RekfDev d)
{
    __shared__ double rowbuf[2][REKF_MR_PAD];
    __shared__ double colbuf[2][REKF_MR_PAD];
    const int m = d.ctl->m;
    if (m == 0) return;
    if (m <= 32) solve_body<1>(d, m, rowbuf, colbuf);
    else if (m <= 64) solve_body<2>(d, m, rowbuf, colbuf);
    else if (m <= 96) solve_body<3>(d, m, rowbuf, colbuf);
    else solve_body<4>(d, m, rowbuf, colbuf);
}

// ----------------------------------------------------------------------------
// k_gain: Kn(i, j) = -sum_k W(i,k) Sinv(k,j) with v_mfma_f64_16x16x4_f64,
// computed transposed (MFMA rows <-> j, MFMA cols <-> i) so that the 16 lanes
// of a row group store 128 contiguous bytes of a Kn column.  mu += W y.
// One workgroup = 16 state rows; wave w takes column tiles w, w+4, ...
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gain(RekfDev d)
{
    const RekfCtl *ctl = d.ctl;
    const int m = ctl->m;
    if (m == 0) return;
    const int n = ctl->n, m_pad = ctl->m_pad;
    const int i0 = blockIdx.x * 16;
    if (i0 >= n) return;
    const size_t ld = (size_t)d.ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = lane & 15, kq = lane >> 4;
    for (int jt = wave; jt < m_pad / 16; jt += 4) {
        const int j0 = 16 * jt;
        v4d acc = {0, 0, 0, 0};
        for (int kk = 0; kk < m_pad / 4; ++kk) {
            const int k = 4 * kk + kq;
            const double a = d.Sinv[k + (size_t)(j0 + idx) * REKF_MR_PAD];   // A[j][k] = Sinv(k, j)
            const double b = d.W[(i0 + idx) + (size_t)k * ld];               // B[k][i] = W(i, k)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + kq + 4 * r;                                   // D row
            d.Kn[(i0 + idx) + (size_t)j * ld] = -acc[r];                     // D col = idx
        }
    }
    if (wave == 0 && lane < 16) {
        const int i = i0 + lane;
        if (i < n) {
            double acc = 0;
            for (int k = 0; k < m; ++k) acc += d.W[i + (size_t)k * ld] * d.y[k];   // cc:306
            double v = d.mu[i] + acc;
            if (i == 2) v = atan2(sin(v), cos(v));                                   // cc:307
            d.mu[i] = v;
        }
    }
}

// ----------------------------------------------------------------------------
// k_downdate: P(i,j) += sum_k Kn(i,k) HPt(j,k)   (P <- P - K (H P), cc:308)
//
// 64x64 tile of P per 256-thread workgroup; the Kn row panel and HPt row panel of
// the tile are staged once in LDS ([k][64] doubles each, 32 KiB + 32 KiB) and
// every wave owns a 32x32 sub-tile = 2x2 v_mfma_f64_16x16x4_f64 accumulators
// initialised with P itself, so P is read once and written once.
// The MFMA is evaluated transposed (MFMA M <-> j, N <-> i) and MFMA tile t of a
// pair covers the interleaved rows i = base + 2*idx + t, so that each lane
// holds two adjacent rows of one column: 16-byte global accesses, 256 contiguous
// bytes per 16 lanes, and ONE ds_read_b128 per operand per k-step feeds both
// tiles, conflict-free on the linear [k][64] LDS image.
// ----------------------------------------------------------------------------
#define DT 64
__global__ __launch_bounds__(256, 2) void k_downdate(RekfDev d)
{
    __shared__ __attribute__((aligned(16))) double sK[DT * 64];
    __shared__ __attribute__((aligned(16))) double sW[DT * 64];
    const RekfCtl *ctl = d.ctl;
    const int m_pad = ctl->m_pad;
    if (ctl->m == 0) return;
    const int n = ctl->n;
    const int T = (n + DT - 1) / DT;
    const int I = blockIdx.x, J = blockIdx.y;
    if (I >= T || J >= T) return;
    const size_t ld = (size_t)d.ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int idx = lane & 15, kq = lane >> 4;
    const int wi = wave & 1, wj = wave >> 1;
    const int ib = DT * I + 32 * wi, jb = DT * J + 32 * wj;

    // accumulators <- P sub-tile
    v4d acc[2][2];
    double *Pw = d.P + (size_t)(ib + 2 * idx);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = jb + 2 * (kq + 4 * r) + mt;
            const v2d v = *(const v2d *)(Pw + (size_t)j * ld);
            acc[mt][0][r] = v.x;
            acc[mt][1][r] = v.y;
        }

    for (int k0 = 0; k0 < m_pad; k0 += 64) {
        const int kmax = (m_pad - k0 < 64) ? (m_pad - k0) : 64;
        if (k0 > 0) __syncthreads();
        for (int e = tid; e < kmax * 32; e += 256) {
            const int k = e >> 5, pr = e & 31;
            const v2d kv = *(const v2d *)(d.Kn + (size_t)(DT * I + 2 * pr) + (size_t)(k0 + k) * ld);
            const v2d wv = *(const v2d *)(d.HPt + (size_t)(DT * J + 2 * pr) + (size_t)(k0 + k) * ld);
            *(v2d *)(sK + k * 64 + 2 * pr) = kv;
            *(v2d *)(sW + k * 64 + 2 * pr) = wv;
        }
        __syncthreads();
        for (int kk = 0; kk < kmax / 4; ++kk) {
            const int k = 4 * kk + kq;
            const v2d a2 = *(const v2d *)(sW + k * 64 + 32 * wj + 2 * idx);   // A[j][k] = HP(k,j)
            const v2d b2 = *(const v2d *)(sK + k * 64 + 32 * wi + 2 * idx);   // B[k][i] = Kn(i,k)
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.x, b2.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.x, b2.y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.y, b2.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.y, b2.y, acc[1][1], 0, 0, 0);
        }
    }

#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = jb + 2 * (kq + 4 * r) + mt;
            v2d v;
            v.x = acc[mt][0][r];
            v.y = acc[mt][1][r];
            *(v2d *)(Pw + (size_t)j * ld) = v;
        }
}

// ----------------------------------------------------------------------------
// k_augment (cc:311-364): one workgroup; runs only while the map is growing.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_augment(RekfDev d, RekfFrontArgs A)
{
    __shared__ double Gp[REKF_MAX_OBS_DEV][6];
    __shared__ double Sxi[9];
    __shared__ double RQR[4];
    RekfCtl *ctl = d.ctl;
    const int N2 = ctl->n_new;
    if (N2 == 0) return;
    const int n = ctl->n;
    const int tid = threadIdx.x;
    const size_t ld = (size_t)d.ld;
    double *P = d.P;
    {
#pragma clang fp contract(off)
        const double x = d.mu[0], y = d.mu[1], th = d.mu[2];
        const double s = sin(th), c = cos(th);                      // cc:323-324
        if (tid < N2) {
            const int local_id = ctl->new_ids[tid];                 // cc:338
            float gx, gy;
            obs_to_global(x, y, th, A.obs[2 * local_id], A.obs[2 * local_id + 1], gx, gy);
            d.mu[n + 2 * tid] = (double)gx;                         // cc:341-342 (float32-rounded)
            d.mu[n + 2 * tid + 1] = (double)gy;
            const double rx = (double)A.obs[2 * local_id], ry = (double)A.obs[2 * local_id + 1];
            Gp[tid][0] = 1.; Gp[tid][1] = 0.; Gp[tid][2] = -rx * s - ry * c;   // cc:347
            Gp[tid][3] = 0.; Gp[tid][4] = 1.; Gp[tid][5] = rx * c - ry * s;
        }
        if (tid == 0) {
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) Sxi[i * 3 + j] = P[i + (size_t)j * ld];   // cc:322
            const double q = A.obs_cov;                             // Gz Qt Gz^T, Gz = R(theta) (cc:326,354)
            RQR[0] = c * q * c + (-s) * q * (-s); RQR[1] = c * q * s + (-s) * q * c;
            RQR[2] = s * q * c + c * q * (-s);    RQR[3] = s * q * s + c * q * c;
        }
    }
    __syncthreads();
    // sigma_mx = G_fx * sigma (cc:355-357): rows n+2a+rr, all old columns, and the mirror
    for (int e = tid; e < n * N2; e += 1024) {
#pragma clang fp contract(off)
        const int a = e / n, col = e - a * n;
        const double q0 = P[0 + (size_t)col * ld], q1 = P[1 + (size_t)col * ld], q2 = P[2 + (size_t)col * ld];
        for (int rr = 0; rr < 2; ++rr) {
            double acc = 0;
            acc += Gp[a][rr * 3 + 0] * q0;
            acc += Gp[a][rr * 3 + 1] * q1;
            acc += Gp[a][rr * 3 + 2] * q2;
            P[(size_t)(n + 2 * a + rr) + (size_t)col * ld] = acc;
            P[(size_t)col + (size_t)(n + 2 * a + rr) * ld] = acc;
        }
    }
    // sigma_mm (cc:354,358): every (a,b) block, a != b included, gets + R Qt R^T
    for (int e = tid; e < N2 * N2; e += 1024) {
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
                P[(size_t)(n + 2 * a + rr) + (size_t)(n + 2 * b + cc) * ld] = acc + RQR[rr * 2 + cc];
            }
    }
    __syncthreads();
    if (tid == 0) ctl->n = n + 2 * N2;                              // cc:360-363
}

// ----------------------------------------------------------------------------
// PredictState, pose block (cc:97-152): non-mutating; out = mu3 | sigma3x3 col-major
// ----------------------------------------------------------------------------
__global__ void k_predict_pose(RekfDev d, RekfFrontArgs A, double *out12)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Motion mo;
    motion_terms(A, d.mu[2], mo);
    double C[9];
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) C[i + 3 * j] = d.P[i + (size_t)j * d.ld];
    corner_predict(C, 3, mo);
    double th = d.mu[2] + mo.d[2];
    th = atan2(sin(th), cos(th));
    out12[0] = d.mu[0] + mo.d[0];
    out12[1] = d.mu[1] + mo.d[1];
    out12[2] = th;
    for (int q = 0; q < 9; ++q) out12[3 + q] = C[q];
}

// ----------------------------------------------------------------------------
// launch wrappers
// ----------------------------------------------------------------------------
void rekf_launch_front(const RekfDev &d, const RekfFrontArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(k_front, dim3(1), dim3(1024), 0, s, d, a);
}
void rekf_launch_gather(const RekfDev &d, int n_ub, hipStream_t s)
{
    hipLaunchKernelGGL(k_gather, dim3((n_ub + 255) / 256, 8), dim3(256), 0, s, d);
}
void rekf_launch_solve(const RekfDev &d, hipStream_t s)
{
    hipLaunchKernelGGL(k_solve, dim3(1), dim3(1024), 0, s, d);
}
void rekf_launch_gain(const RekfDev &d, int n_ub, hipStream_t s)
{
    hipLaunchKernelGGL(k_gain, dim3((n_ub + 15) / 16), dim3(256), 0, s, d);
}
void rekf_launch_downdate(const RekfDev &d, int n_ub, hipStream_t s)
{
    const int T = (n_ub + DT - 1) / DT;
    hipLaunchKernelGGL(k_downdate, dim3(T, T), dim3(256), 0, s, d);
}
void rekf_launch_augment(const RekfDev &d, const RekfFrontArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(k_augment, dim3(1), dim3(1024), 0, s, d, a);
}
void rekf_launch_predict_pose(const RekfDev &d, const RekfFrontArgs &a, double *out12, hipStream_t s)
{
    hipLaunchKernelGGL(k_predict_pose, dim3(1), dim3(64), 0, s, d, a, out12);
}
