/*
 * oracle/ekf_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, FP64, single thread) of the reference's EKF-SLAM
 * core, src/reflector_ekf_slam/reflector_ekf_slam.cc, and of the extra
 * pose-fusion branch of src/reflector_ekf_slam/reflector_ekf_slam_gps.cc.
 * All file:line citations below are into /root/reference.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
 * for this path and cannot be compiled on this image (it needs Eigen, glog
 * and ROS headers that are neither vendored nor installed), so this
 * restatement is pinned only by (1) an independent numpy restatement
 * (oracle/ekf_numpy.py) that follows the Eigen expressions literally,
 * (2) literal-mode == structured-mode, and (3) hand-checkable micro cases in
 * tests/.  See DESIGN.md "Oracle".
 *
 * Two execution modes with the same mathematics:
 *   literal    - performs the dense operation sequence the Eigen expressions
 *                execute (dense G P G^T, gain expression evaluated twice,
 *                (K H) P association): the "reference CPU path" for timing.
 *   structured - exploits the sparsity of G and H (O(n^2 m)); what a fair CPU
 *                implementation would do and what the HIP path implements.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (reflector_ekf_slam_amd/) never does.
 *
 * Compile this file with -ffp-contract=off: the reference is built without
 * FMA (CMakeLists.txt:4-6, plain -O3 x86-64), and the float32 roundings in
 * association/augment make single-rounding differences observable.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

void od_gemm_acc(int M, int N, int K, double alpha, const double *A, int lda,
                 const double *B, int ldb, double *C, int ldc);
void od_gemm(int M, int N, int K, const double *A, int lda, const double *B,
             int ldb, double *C, int ldc);
void od_transpose(int M, int N, const double *A, int lda, double *B, int ldb);
int od_lu_inverse(int m, double *A, int lda);
int od_get_threads(void);
void od_gemm_acc_block(int M, int N, int K, double alpha, const double *A, int lda, const double *B, int ldb, double *C, int ldc);
#ifdef _OPENMP
#include <omp.h>
#endif

enum { OEKF_DIFF = 0, OEKF_OMNI = 1 };

typedef struct oekf {
    int model;        /* sensor::OdometryModel (sensor_data.h:56-60) */
    int literal;      /* 0 structured, 1 literal */
    double time;      /* State::time   (ekf_slam_interface.h:43-48) */
    int n;            /* rows of mu */
    double *mu;       /* State::mu */
    double *P;        /* State::sigma, column-major, ld = n (like Eigen::MatrixXd) */
    double vt[3];     /* vt_ (reflector_ekf_slam.h:52) */
    double lin_cov, ang_cov, obs_cov; /* EKFOptions (ekf_slam_interface.h:36-40) */
    /* pre-loaded map (sensor_data.h:30-37) */
    int M_map;
    float *map_xy;    /* M_map x 2 float32 */
    double *map_cov;  /* M_map x 4, row-major 2x2 */
    /* last ReflectorMatchResult (ekf_slam_interface.h:18-26) */
    int n_state, n_map, n_new, cap_match;
    int *state_pairs; /* (obs, landmark) */
    int *map_pairs;   /* (obs, map point) */
    int *new_ids;
} oekf_t;

/* ---- ctor: reflector_ekf_slam.cc:6-37 ---------------------------------- */
oekf_t *oekf_create(int odom_model, double init_time, const double init_pose[3],
                    double linear_velocity_cov, double angular_velocity_cov,
                    double observation_cov)
{
    oekf_t *e = (oekf_t *)calloc(1, sizeof(oekf_t));
    e->model = (odom_model == OEKF_DIFF) ? OEKF_DIFF : OEKF_OMNI; /* :13-32 default == OMNI */
    e->time = init_time;                                          /* :8 */
    e->n = 3;
    e->mu = (double *)calloc(3, sizeof(double));
    memcpy(e->mu, init_pose, 3 * sizeof(double));                  /* :9 */
    e->P = (double *)calloc(9, sizeof(double));                    /* :10-11 */
    e->lin_cov = linear_velocity_cov;
    e->ang_cov = angular_velocity_cov;
    e->obs_cov = observation_cov;                                  /* Qt_ = obs*I2, :33-34 */
    return e;
}

void oekf_destroy(oekf_t *e)
{
    if (!e) return;
    free(e->mu); free(e->P); free(e->map_xy); free(e->map_cov);
    free(e->state_pairs); free(e->map_pairs); free(e->new_ids);
    free(e);
}

void oekf_set_mode(oekf_t *e, int literal) { e->literal = literal ? 1 : 0; }

/* map_ as LoadMapFromTxtFile would have filled it (reflector_ekf_slam.cc:80-94;
 * the txt parsing itself is host-side code in the product). */
int oekf_set_map(oekf_t *e, const float *xy, const double *cov, int M)
{
    free(e->map_xy); free(e->map_cov);
    e->map_xy = NULL; e->map_cov = NULL; e->M_map = 0;
    if (M <= 0) return 0;
    e->map_xy = (float *)malloc(sizeof(float) * 2 * (size_t)M);
    e->map_cov = (double *)malloc(sizeof(double) * 4 * (size_t)M);
    memcpy(e->map_xy, xy, sizeof(float) * 2 * (size_t)M);
    memcpy(e->map_cov, cov, sizeof(double) * 4 * (size_t)M);
    e->M_map = M;
    return 0;
}

/* ---- motion model pieces shared by Predict and PredictState -------------
 * reflector_ekf_slam.cc:156-205 (Predict) and :97-152 (PredictState) are the
 * same formulas; out: d[3] mean increment, a,b = G(0,2), G(1,2), V = 3x3
 * row-major Gu Qu Gu^T. */
static void motion_terms(const oekf_t *e, double theta, double dt,
                         double d[3], double *a, double *b, double V[9])
{
    const double vx = e->vt[0], vy = e->vt[1], w = e->vt[2];
    double Gu[9] = {0}; /* 3 x q, row-major, q = 2 (DIFF) or 3 */
    int q;
    double Qu[3];
    if (e->model == OEKF_DIFF) {
        const double delta_theta = w * dt;                        /* :158 */
        const double half = theta + delta_theta / 2;              /* :165 */
        d[0] = vx * dt * cos(half);                               /* :159 */
        d[1] = vx * dt * sin(half);                               /* :160 */
        d[2] = delta_theta;
        *a = -vx * dt * sin(half);                                /* :167 */
        *b = vx * dt * cos(half);                                 /* :168 */
        q = 2;
        Gu[0] = dt * cos(half); Gu[1] = -vx * dt * dt * sin(half) / 2; /* :173 */
        Gu[3] = dt * sin(half); Gu[4] = vx * dt * dt * cos(half) / 2;  /* :174 */
        Gu[6] = 0;              Gu[7] = dt;                             /* :175 */
        Qu[0] = e->lin_cov; Qu[1] = e->ang_cov;                   /* :16-18 */
    } else {
        const double delta_theta = w * dt;                        /* :184 */
        d[0] = vx * dt * cos(theta) - vy * dt * sin(theta);       /* :185 */
        d[1] = vx * dt * sin(theta) + vy * dt * cos(theta);       /* :186 */
        d[2] = delta_theta;
        *a = -vx * dt * sin(theta) - vy * dt * cos(theta);        /* :191 */
        *b = vx * dt * cos(theta) - vy * dt * sin(theta);         /* :192 */
        q = 3;
        Gu[0] = dt * cos(theta); Gu[1] = -dt * sin(theta); Gu[2] = 0.; /* :197 */
        Gu[3] = dt * sin(theta); Gu[4] = dt * cos(theta);  Gu[5] = 0.; /* :198 */
        Gu[6] = 0.;              Gu[7] = 0.;               Gu[8] = dt; /* :199 */
        Qu[0] = e->lin_cov; Qu[1] = e->lin_cov; Qu[2] = e->ang_cov;    /* :21-24 */
    }
    const int ld = (q == 2) ? 3 : 3; /* Gu stored with row stride 3 */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < q; ++k)
                s += Gu[i * ld + k] * Qu[k] * Gu[j * ld + k];
            V[i * 3 + j] = s;
        }
}

/* P <- G P G^T + Gu Qu Gu^T on an n x n column-major matrix (ld = n). */
static void cov_predict_structured(double *P, int n, double a, double b, const double V[9])
{
    /* G = I + a e0 e2^T + b e1 e2^T.  (G P): row0 += a row2, row1 += b row2. */
    for (int c = 0; c < n; ++c) {
        const double p2 = P[2 + (size_t)c * n];
        P[0 + (size_t)c * n] = P[0 + (size_t)c * n] + a * p2;
        P[1 + (size_t)c * n] = P[1 + (size_t)c * n] + b * p2;
    }
    /* (.) G^T: col0 += a col2, col1 += b col2 (using the updated rows). */
    for (int r = 0; r < n; ++r) {
        const double p2 = P[r + (size_t)2 * n];
        P[r + (size_t)0 * n] = P[r + (size_t)0 * n] + a * p2;
        P[r + (size_t)1 * n] = P[r + (size_t)1 * n] + b * p2;
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            P[i + (size_t)j * n] += V[i * 3 + j];
}

static void cov_predict_literal(double *P, int n, double a, double b, const double V[9])
{
    /* Dense G_xi * sigma * G_xi^T as written at reflector_ekf_slam.cc:178/:202. */
    double *G = (double *)calloc((size_t)n * n, sizeof(double));
    double *Gt = (double *)calloc((size_t)n * n, sizeof(double));
    double *T = (double *)malloc(sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; ++i) { G[i + (size_t)i * n] = 1.0; Gt[i + (size_t)i * n] = 1.0; }
    G[0 + (size_t)2 * n] = a;  G[1 + (size_t)2 * n] = b;
    Gt[2 + (size_t)0 * n] = a; Gt[2 + (size_t)1 * n] = b;
    od_gemm(n, n, n, G, n, P, n, T, n);
    od_gemm(n, n, n, T, n, Gt, n, P, n);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            P[i + (size_t)j * n] += V[i * 3 + j];
    free(G); free(Gt); free(T);
}

/* ---- Predict: reflector_ekf_slam.cc:154-206 ------------------------------ */
static void predict(oekf_t *e, double dt)
{
    double d[3], a, b, V[9];
    motion_terms(e, e->mu[2], dt, d, &a, &b, V);
    if (e->literal) cov_predict_literal(e->P, e->n, a, b, V);
    else cov_predict_structured(e->P, e->n, a, b, V);
    e->mu[0] += d[0]; e->mu[1] += d[1]; e->mu[2] += d[2];          /* :180/:204 */
    e->mu[2] = atan2(sin(e->mu[2]), cos(e->mu[2]));                 /* :181/:205 */
}

/* ---- PredictState: reflector_ekf_slam.cc:97-152 (non-mutating) ----------- */
void oekf_predict_state(const oekf_t *e, double time, double *mu_out, double *sigma_out)
{
    const int n = e->n;
    const double dt = time - e->time;                                /* :100 */
    double d[3], a, b, V[9];
    motion_terms(e, e->mu[2], dt, d, &a, &b, V);
    memcpy(mu_out, e->mu, sizeof(double) * (size_t)n);
    mu_out[0] = e->mu[0] + d[0]; mu_out[1] = e->mu[1] + d[1]; mu_out[2] = e->mu[2] + d[2];
    mu_out[2] = atan2(sin(mu_out[2]), cos(mu_out[2]));               /* :126/:150 */
    if (sigma_out) {
        memcpy(sigma_out, e->P, sizeof(double) * (size_t)n * n);
        cov_predict_structured(sigma_out, n, a, b, V);
    }
}

/* ---- HandleOdometryMessage: reflector_ekf_slam.cc:208-223 ---------------- */
void oekf_handle_odometry(oekf_t *e, double t, double vx, double vy, double wz)
{
    if (t < e->time) return;                                         /* :211-212 */
    e->vt[0] = vx; e->vt[1] = vy; e->vt[2] = wz;                     /* :216 */
    predict(e, t - e->time);                                         /* :217-218 */
    e->time = t;                                                     /* :219 */
}

/* ---- ReflectorMatch: reflector_ekf_slam.cc:370-455 ----------------------- */
static void ensure_match_cap(oekf_t *e, int K)
{
    if (K <= e->cap_match) return;
    e->state_pairs = (int *)realloc(e->state_pairs, sizeof(int) * 2 * (size_t)K);
    e->map_pairs = (int *)realloc(e->map_pairs, sizeof(int) * 2 * (size_t)K);
    e->new_ids = (int *)realloc(e->new_ids, sizeof(int) * (size_t)K);
    e->cap_match = K;
}

static __attribute__((noinline)) void obs_to_global(const double *mu, float px, float py, float *gx, float *gy)
{
    /* :389-393 / :327-331: float * double promotes to double; the result is
     * rounded to float32 on assignment. */
    const float x = (float)((double)px * cos(mu[2]) - (double)py * sin(mu[2]) + mu[0]);
    const float y = (float)((double)px * sin(mu[2]) + (double)py * cos(mu[2]) + mu[1]);
    *gx = x; *gy = y;
}

static void reflector_match_team(oekf_t *e, const float *obs, int K, int T);
static void reflector_match(oekf_t *e, const float *obs, int K)
{
    ensure_match_cap(e, K);
    e->n_state = e->n_map = e->n_new = 0;
    if (e->n == 3 && e->M_map == 0) {                                /* :379-387 */
        for (int i = 0; i < K; ++i) e->new_ids[e->n_new++] = i;
        return;
    }
    if (!e->literal && od_get_threads() > 1 && K > 1) { reflector_match_team(e, obs, K, od_get_threads()); return; }
    const int M = (e->n - 3) / 2;                                    /* :395 */
    const int M_ = e->M_map;                                         /* :396 */
    for (int i = 0; i < K; ++i) {
        float gx, gy;
        obs_to_global(e->mu, obs[2 * i], obs[2 * i + 1], &gx, &gy);  /* :399 */
        if (M_ > 0) {                                                /* :401-425 */
            double best = 0; int bj = -1;
            for (int j = 0; j < M_; ++j) {
                const double *S = e->map_cov + 4 * (size_t)j;        /* :407 */
                const float ex = e->map_xy[2 * j] - gx;              /* :408 float32 subtract */
                const float ey = e->map_xy[2 * j + 1] - gy;
                const double dx = (double)ex, dy = (double)ey;       /* :409 */
                /* :411 delta * sigma * delta^T: (1x2 * 2x2) then * 2x1 */
                const double t0 = dx * S[0] + dy * S[2];
                const double t1 = dx * S[1] + dy * S[3];
                const double dist = sqrt(t0 * dx + t1 * dy);
                /* :414-419 sort with '<=' then front(): first minimum (ties are UB
                 * in the reference; NaN never wins here). */
                if (bj < 0 || dist < best) { best = dist; bj = j; }
            }
            if (best < 0.05) {                                       /* :420 */
                e->map_pairs[2 * e->n_map] = i;
                e->map_pairs[2 * e->n_map + 1] = bj;
                e->n_map++;
                continue;
            }
        }
        if (M > 0) {                                                 /* :426-451 */
            double best = 0; int bj = -1;
            for (int j = 0; j < M; ++j) {
                const float lx = (float)e->mu[3 + 2 * j];            /* :431 Vector2f(double,double) */
                const float ly = (float)e->mu[3 + 2 * j + 1];
                const float ex = gx - lx;                            /* :433 */
                const float ey = gy - ly;
                const double dx = (double)ex, dy = (double)ey;       /* :434 */
                const double dist = sqrt(dx * dx + dy * dy);         /* :437 */
                if (bj < 0 || dist < best) { best = dist; bj = j; }
            }
            if (best < 0.6) {                                        /* :446 */
                e->state_pairs[2 * e->n_state] = i;
                e->state_pairs[2 * e->n_state + 1] = bj;
                e->n_state++;
                continue;
            }
        }
        e->new_ids[e->n_new++] = i;                                  /* :452 */
    }
}

/* The same decisions (cc:397-453), one observation per loop iteration with the iterations dealt to an OpenMP team (the structured
 * all-core baseline): every observation's result depends only on the state, so the per-observation loop bodies are the
 * ones above, verbatim; the three lists are filled in observation order behind the loop. */
static void reflector_match_team(oekf_t *e, const float *obs, int K, int T)
{
    const int M = (e->n - 3) / 2, M_ = e->M_map;
    int *kind = (int *)malloc(sizeof(int) * 2 * (size_t)K), *idx = kind + K;
#pragma omp parallel for schedule(static) num_threads(T)
    for (int i = 0; i < K; ++i) {
        float gx, gy;
        obs_to_global(e->mu, obs[2 * i], obs[2 * i + 1], &gx, &gy);  /* :399 */
        kind[i] = 2; idx[i] = -1;
        if (M_ > 0) {                                                /* :401-425 */
            double best = 0; int bj = -1;
            for (int j = 0; j < M_; ++j) {
                const double *S = e->map_cov + 4 * (size_t)j;
                const float ex = e->map_xy[2 * j] - gx;
                const float ey = e->map_xy[2 * j + 1] - gy;
                const double dx = (double)ex, dy = (double)ey;
                const double t0 = dx * S[0] + dy * S[2];
                const double t1 = dx * S[1] + dy * S[3];
                const double dist = sqrt(t0 * dx + t1 * dy);
                if (bj < 0 || dist < best) { best = dist; bj = j; }
            }
            if (best < 0.05) { kind[i] = 0; idx[i] = bj; continue; } /* :420 */
        }
        if (M > 0) {                                                 /* :426-451 */
            double best = 0; int bj = -1;
            for (int j = 0; j < M; ++j) {
                const float lx = (float)e->mu[3 + 2 * j];
                const float ly = (float)e->mu[3 + 2 * j + 1];
                const float ex = gx - lx;
                const float ey = gy - ly;
                const double dx = (double)ex, dy = (double)ey;
                const double dist = sqrt(dx * dx + dy * dy);
                if (bj < 0 || dist < best) { best = dist; bj = j; }
            }
            if (best < 0.6) { kind[i] = 1; idx[i] = bj; }            /* :446 */
        }
    }
    for (int i = 0; i < K; ++i) {
        if (kind[i] == 0) { e->map_pairs[2 * e->n_map] = i; e->map_pairs[2 * e->n_map + 1] = idx[i]; e->n_map++; }
        else if (kind[i] == 1) { e->state_pairs[2 * e->n_state] = i; e->state_pairs[2 * e->n_state + 1] = idx[i]; e->n_state++; }
        else e->new_ids[e->n_new++] = i;                             /* :452 */
    }
    free(kind);
}

/* The structured update (cc:246-309 with the sparsity of H exploited: O(n^2 m)) by ONE persistent OpenMP team per update -- the fair
 * algorithmic CPU baseline SURVEY.md 8(d) asks for.  Thread t OWNS a fixed, contiguous range of P's columns (chunks of 16, the dense
 * kernel's unit) and the same range of rows of the n x m panels: its share of P stays in its own cache from update to update, every
 * phase is a loop over owned rows or columns, and the phases meet at four barriers instead of a fork-join per loop.  Every element sees
 * exactly the operations, in the order, of the single-thread structured path below (same source expressions in this file, the same dense
 * kernel on column / row sub-blocks): bit-identical results at any thread count (tests/test_oracle_cpu.py).
 * hc / hv / hn: the <= 5 structural non-zeros of each H row; returns 0, or -3 when S is singular. */
static int update_structured_team(oekf_t *e, int m, const int *hc, const double *hv, const int *hn, const double *Qd, const double *dz, int T)
{
    const int N = e->n;
    double *PHt = (double *)malloc(sizeof(double) * (size_t)N * m);
    double *Kt = (double *)malloc(sizeof(double) * (size_t)N * m);
    double *HP = (double *)malloc(sizeof(double) * (size_t)m * N);
    double *S = (double *)malloc(sizeof(double) * (size_t)m * m);
    int bad = 0;
    const int chunks = (N + 15) / 16;
#pragma omp parallel num_threads(T)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        const int t = 0, nt = 1;
#endif
        const int per = (chunks + nt - 1) / nt;
        int o0 = 16 * per * t, o1 = 16 * per * (t + 1);                /* owned rows of the panels = owned columns of P */
        if (o0 > N) o0 = N;
        if (o1 > N) o1 = N;
        /* (P H^T)(r, j) = sum over the <= 5 non-zeros of H row j, owned rows */
        for (int j = 0; j < m; ++j) {
            double *w = PHt + (size_t)j * N;
            for (int r = o0; r < o1; ++r) w[r] = 0.0;
            for (int q = 0; q < hn[j]; ++q) {
                const double h = hv[5 * j + q];
                const double *pc = e->P + (size_t)hc[5 * j + q] * N;
                for (int r = o0; r < o1; ++r) w[r] += pc[r] * h;
            }
        }
        /* (H P)(i, c) from the rows of P, owned columns (read before anybody writes P: the barrier below) */
        for (int cidx = o0; cidx < o1; ++cidx) {
            const double *pc = e->P + (size_t)cidx * N;
            for (int i = 0; i < m; ++i) {
                double acc = 0;
                for (int q = 0; q < hn[i]; ++q)
                    acc += hv[5 * i + q] * pc[hc[5 * i + q]];
                HP[i + (size_t)cidx * m] = acc;
            }
        }
#pragma omp barrier
#pragma omp single
        {
            for (int j = 0; j < m; ++j)
                for (int i = 0; i < m; ++i) {
                    double acc = 0;
                    for (int q = 0; q < hn[i]; ++q)
                        acc += hv[5 * i + q] * PHt[hc[5 * i + q] + (size_t)j * N];
                    S[i + (size_t)j * m] = acc;
                }
            for (int r = 0; r < m; ++r) S[r + (size_t)r * m] += Qd[r];
            if (od_lu_inverse(m, S, m) != 0) bad = 1;
        }                                                              /* (implicit barrier) */
        if (!bad && o1 > o0) {
            /* K_t rows: the dense kernel on this thread's row block (od_gemm = zero + accumulate) */
            for (int j = 0; j < m; ++j)
                memset(Kt + o0 + (size_t)j * N, 0, sizeof(double) * (size_t)(o1 - o0));
            od_gemm_acc_block(o1 - o0, m, m, 1.0, PHt + o0, N, S, m, Kt + o0, N);
            for (int r = o0; r < o1; ++r) {                            /* :306 */
                double acc = 0;
                for (int j = 0; j < m; ++j) acc += Kt[r + (size_t)j * N] * dz[j];
                e->mu[r] += acc;
            }
        }
#pragma omp barrier
        if (!bad) {
            /* P -= K (H P), owned columns, 16 at a time (the dense kernel's column grouping is part of the arithmetic) */
            for (int j0 = o0; j0 < o1; j0 += 16) {
                const int nj = (o1 - j0 < 16) ? (o1 - j0) : 16;
                od_gemm_acc_block(N, nj, m, -1.0, Kt, N, HP + (size_t)j0 * m, m, e->P + (size_t)j0 * N, N);
            }
        }
    }
    if (!bad) e->mu[2] = atan2(sin(e->mu[2]), cos(e->mu[2]));          /* :307 */
    free(PHt); free(Kt); free(HP); free(S);
    return bad ? -3 : 0;
}

/* quaternion (w,0,0,z) -> angle-axis z component: transform.h:46-70 as used at
 * reflector_ekf_slam_gps.cc:320-322. */
static double yaw_innovation(double delta_theta)
{
    double w = cos(delta_theta / 2), z = sin(delta_theta / 2);
    const double nrm = sqrt(w * w + z * z);
    w /= nrm; z /= nrm;
    if (w < 0.) { w = -w; z = -z; }
    const double angle = 2. * atan2(fabs(z), w);
    const double scale = angle < 1e-7 ? 2. : angle / sin(angle / 2.);
    return scale * z;
}

/* ---- HandleObservationMessage: reflector_ekf_slam.cc:229-368,
 *      pose-fusion branch: reflector_ekf_slam_gps.cc:305-340 ---------------- */
int oekf_handle_observation(oekf_t *e, double t, const float *obs, int K,
                            const double *gps_pose3 /* nullable: x, y, yaw */)
{
    predict(e, t - e->time);                                         /* :232-233 */
    e->time = t;                                                     /* :234 */
    e->n_state = e->n_map = e->n_new = 0;
    if (K <= 0) return 0;                                            /* :235-236 */
    reflector_match(e, obs, K);                                      /* :237 */
    const int M = e->n_state, M_ = e->n_map, MM = M + M_;
    const int N = e->n;
    if (MM > 0) {                                                    /* :246 */
        const int m = 2 * MM + (gps_pose3 ? 3 : 0);
        /* the dense H (m x N col-major) only where the dense products need it: the structured mode works from its <= 5 non-zeros per row */
        double *H = (double *)calloc(e->literal ? (size_t)m * N : (size_t)m * 5, sizeof(double));
        const int ldh = m;
#define HSET(r, col, slot, v) do { if (e->literal) H[(r) + (size_t)(col) * ldh] = (v); else H[(r) + (size_t)(slot) * ldh] = (v); } while (0)
        double *z = (double *)calloc((size_t)m, sizeof(double));
        double *zh = (double *)calloc((size_t)m, sizeof(double));
        double *Qd = (double *)calloc((size_t)m, sizeof(double));     /* diagonal of Q */
        /* the <=5 structural nonzeros of each H row: columns 0,1,2 and the landmark pair */
        int *hc = (int *)calloc((size_t)m * 5, sizeof(int));
        double *hv = (double *)calloc((size_t)m * 5, sizeof(double));
        int *hn = (int *)calloc((size_t)m, sizeof(int));
        const double c = cos(e->mu[2]), s = sin(e->mu[2]);            /* :252-253 */
        for (int i = 0; i < MM; ++i) {
            const int is_state = i < M;
            const int local_id = is_state ? e->state_pairs[2 * i] : e->map_pairs[2 * (i - M)];
            const int global_id = is_state ? e->state_pairs[2 * i + 1] : e->map_pairs[2 * (i - M) + 1];
            z[2 * i] = (double)obs[2 * local_id];                     /* :265-266 / :289-290 */
            z[2 * i + 1] = (double)obs[2 * local_id + 1];
            double lx, ly;
            if (is_state) { lx = e->mu[3 + 2 * global_id]; ly = e->mu[3 + 2 * global_id + 1]; } /* :259 */
            else { lx = (double)e->map_xy[2 * global_id]; ly = (double)e->map_xy[2 * global_id + 1]; } /* :282 */
            const double dx = lx - e->mu[0];                          /* :267 / :291 */
            const double dy = ly - e->mu[1];
            zh[2 * i] = dx * c + dy * s;                              /* :269 */
            zh[2 * i + 1] = -dx * s + dy * c;                         /* :270 */
            /* A_i (:272-273) */
            HSET(2 * i, 0, 0, -c);
            HSET(2 * i, 1, 1, -s);
            HSET(2 * i, 2, 2, -dx * s + dy * c);
            HSET(2 * i + 1, 0, 0, s);
            HSET(2 * i + 1, 1, 1, -c);
            HSET(2 * i + 1, 2, 2, -dx * c - dy * s);
            if (is_state) {                                           /* B (:255,:275); map rows have none (:300) */
                const int col = 3 + 2 * global_id;
                HSET(2 * i, col, 3, c);
                HSET(2 * i, col + 1, 4, s);
                HSET(2 * i + 1, col, 3, -s);
                HSET(2 * i + 1, col + 1, 4, c);
            }
            Qd[2 * i] = e->obs_cov; Qd[2 * i + 1] = e->obs_cov;       /* :276 / :302 */
        }
        double *dz = (double *)malloc(sizeof(double) * (size_t)m);
        for (int r = 0; r < 2 * MM; ++r) dz[r] = z[r] - zh[r];
        if (gps_pose3) {                                              /* gps.cc:305-332 */
            const int r0 = 2 * MM;
            for (int k = 0; k < 3; ++k) HSET(r0 + k, k, k, 1.0);
            dz[r0] = gps_pose3[0] - e->mu[0];
            dz[r0 + 1] = gps_pose3[1] - e->mu[1];
            dz[r0 + 2] = yaw_innovation(gps_pose3[2] - e->mu[2]);
            Qd[r0] = 0.05 * 0.05; Qd[r0 + 1] = 0.05 * 0.05; Qd[r0 + 2] = 0.017 * 0.017;
        }

        for (int r = 0; r < m; ++r) {
            const int cand[5] = {0, 1, 2,
                                 (r < 2 * M) ? 3 + 2 * e->state_pairs[2 * (r / 2) + 1] : -1,
                                 (r < 2 * M) ? 4 + 2 * e->state_pairs[2 * (r / 2) + 1] : -1};
            for (int q = 0; q < 5; ++q)
                if (cand[q] >= 0) {
                    hc[5 * r + hn[r]] = cand[q];
                    hv[5 * r + hn[r]] = e->literal ? H[r + (size_t)cand[q] * m] : H[r + (size_t)q * ldh];
                    hn[r]++;
                }
        }
#undef HSET
        if (!e->literal && od_get_threads() > 1) {                   /* the all-core baseline: one persistent team per update */
            const int rct = update_structured_team(e, m, hc, hv, hn, Qd, dz, od_get_threads());
            free(H); free(z); free(zh); free(Qd); free(dz);
            free(hc); free(hv); free(hn);
            if (rct != 0) return rct;
            goto update_done;
        }
        double *Ht = (double *)malloc(sizeof(double) * (size_t)N * m); /* N x m */
        double *PHt = (double *)malloc(sizeof(double) * (size_t)N * m);
        double *S = (double *)malloc(sizeof(double) * (size_t)m * m);
        double *Kt = (double *)malloc(sizeof(double) * (size_t)N * m);
        if (e->literal) od_transpose(m, N, H, m, Ht, N);
        const int passes = e->literal ? 2 : 1; /* lazy 'auto K_t' evaluated at :306 and :308 */
        for (int pass = 0; pass < passes; ++pass) {
            if (e->literal) {
                od_gemm(N, m, N, e->P, N, Ht, N, PHt, N);             /* sigma * H^T */
                od_gemm(m, m, N, H, m, PHt, N, S, m);                 /* H * (sigma H^T) */
            } else {
                /* column gather: (P H^T)(r, j) = sum over the <=5 nonzeros of H row j */
                for (int j = 0; j < m; ++j) {
                    double *w = PHt + (size_t)j * N;
                    memset(w, 0, sizeof(double) * (size_t)N);
                    for (int q = 0; q < hn[j]; ++q) {
                        const double h = hv[5 * j + q];
                        const double *pc = e->P + (size_t)hc[5 * j + q] * N;
                        for (int r = 0; r < N; ++r) w[r] += pc[r] * h;
                    }
                }
                for (int j = 0; j < m; ++j)
                    for (int i = 0; i < m; ++i) {
                        double acc = 0;
                        for (int q = 0; q < hn[i]; ++q)
                            acc += hv[5 * i + q] * PHt[hc[5 * i + q] + (size_t)j * N];
                        S[i + (size_t)j * m] = acc;
                    }
            }
            for (int r = 0; r < m; ++r) S[r + (size_t)r * m] += Qd[r];
            if (od_lu_inverse(m, S, m) != 0) {
                free(H); free(z); free(zh); free(Qd); free(dz);
                free(hc); free(hv); free(hn);
                free(Ht); free(PHt); free(S); free(Kt);
                return -3;
            }
            od_gemm(N, m, m, PHt, N, S, m, Kt, N);                    /* K_t */
            if (pass == 0) {
                for (int r = 0; r < N; ++r) {                          /* :306 */
                    double acc = 0;
                    for (int j = 0; j < m; ++j) acc += Kt[r + (size_t)j * N] * dz[j];
                    e->mu[r] += acc;
                }
                e->mu[2] = atan2(sin(e->mu[2]), cos(e->mu[2]));        /* :307 */
            }
        }
        if (e->literal) {                                             /* :308 (K_t*H_t)*sigma */
            double *KH = (double *)malloc(sizeof(double) * (size_t)N * N);
            double *T = (double *)malloc(sizeof(double) * (size_t)N * N);
            od_gemm(N, N, m, Kt, N, H, m, KH, N);
            od_gemm(N, N, N, KH, N, e->P, N, T, N);
            for (size_t q = 0; q < (size_t)N * N; ++q) e->P[q] -= T[q];
            free(KH); free(T);
        } else {
            /* HP from rows of P, then P -= K * HP */
            double *HP = (double *)malloc(sizeof(double) * (size_t)m * N);
            for (int cidx = 0; cidx < N; ++cidx) {
                const double *pc = e->P + (size_t)cidx * N;
                for (int i = 0; i < m; ++i) {
                    double acc = 0;
                    for (int q = 0; q < hn[i]; ++q)
                        acc += hv[5 * i + q] * pc[hc[5 * i + q]];
                    HP[i + (size_t)cidx * m] = acc;
                }
            }
            od_gemm_acc(N, N, m, -1.0, Kt, N, HP, m, e->P, N);
            free(HP);
        }
        free(H); free(z); free(zh); free(Qd); free(dz);
        free(hc); free(hv); free(hn);
        free(Ht); free(PHt); free(S); free(Kt);
    }
update_done:;

    const int N2 = e->n_new;                                          /* :311 */
    if (N2 > 0) {
        const int Me = N + 2 * N2;                                    /* :316 */
        double *xe = (double *)calloc((size_t)Me, sizeof(double));
        double *Sg = (double *)calloc((size_t)Me * Me, sizeof(double));
        memcpy(xe, e->mu, sizeof(double) * (size_t)N);                /* :318 */
        for (int j = 0; j < N; ++j)                                   /* :321 */
            memcpy(Sg + (size_t)j * Me, e->P + (size_t)j * N, sizeof(double) * (size_t)N);
        double Sxi[9];                                                /* :322, [i][j] */
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Sxi[i * 3 + j] = e->P[i + (size_t)j * N];
        const double s = sin(e->mu[2]), c = cos(e->mu[2]);            /* :323-324 */
        double *Gp = (double *)malloc(sizeof(double) * 6 * (size_t)N2); /* per new: 2x3 row-major */
        for (int i = 0; i < N2; ++i) {
            const int local_id = e->new_ids[i];                       /* :338 */
            float gx, gy;
            obs_to_global(e->mu, obs[2 * local_id], obs[2 * local_id + 1], &gx, &gy); /* :339 */
            xe[N + 2 * i] = (double)gx;                               /* :341-342 */
            xe[N + 2 * i + 1] = (double)gy;
            const double rx = (double)obs[2 * local_id];              /* :344-345 */
            const double ry = (double)obs[2 * local_id + 1];
            double *g = Gp + 6 * (size_t)i;                           /* :347 */
            g[0] = 1.; g[1] = 0.; g[2] = -rx * s - ry * c;
            g[3] = 0.; g[4] = 1.; g[5] = rx * c - ry * s;
        }
        /* Gz Qt Gz^T with Gz = R(theta) for every row pair (:349,:354) */
        const double q = e->obs_cov;
        const double RQR[4] = { c * q * c + (-s) * q * (-s), c * q * s + (-s) * q * c,
                                s * q * c + c * q * (-s),    s * q * s + c * q * c };
        for (int a = 0; a < N2; ++a) {
            const double *ga = Gp + 6 * (size_t)a;
            /* sigma_mx rows (:355): Gp_a * P[0:3, 0:N] */
            for (int col = 0; col < N; ++col) {
                for (int rr = 0; rr < 2; ++rr) {
                    double acc = 0;
                    for (int k = 0; k < 3; ++k) acc += ga[rr * 3 + k] * e->P[k + (size_t)col * N];
                    Sg[(N + 2 * a + rr) + (size_t)col * Me] = acc;    /* :356 */
                    Sg[col + (size_t)(N + 2 * a + rr) * Me] = acc;    /* :357 */
                }
            }
            /* sigma_mm blocks (:354,:358): every (a,b) incl. a != b gets + R Qt R^T */
            for (int b2 = 0; b2 < N2; ++b2) {
                const double *gb = Gp + 6 * (size_t)b2;
                for (int rr = 0; rr < 2; ++rr)
                    for (int cc = 0; cc < 2; ++cc) {
                        double acc = 0;
                        for (int k = 0; k < 3; ++k) {
                            double t = 0;
                            for (int l = 0; l < 3; ++l) t += ga[rr * 3 + l] * Sxi[l * 3 + k];
                            acc += t * gb[cc * 3 + k];
                        }
                        Sg[(N + 2 * a + rr) + (size_t)(N + 2 * b2 + cc) * Me] = acc + RQR[rr * 2 + cc];
                    }
            }
        }
        free(Gp);
        free(e->mu); free(e->P);
        e->mu = xe; e->P = Sg; e->n = Me;                              /* :360-363 */
    }
    return 0;
}

/* ---- getters: reflector_ekf_slam.h:25-44 --------------------------------- */
int oekf_get_n(const oekf_t *e) { return e->n; }
double oekf_get_time(const oekf_t *e) { return e->time; }
void oekf_get_state(const oekf_t *e, double *mu, double *sigma)
{
    if (mu) memcpy(mu, e->mu, sizeof(double) * (size_t)e->n);
    if (sigma) memcpy(sigma, e->P, sizeof(double) * (size_t)e->n * e->n);
}
/* ---- marker ellipses: the caller's per-landmark covariance ellipse, src/ros_node.cc:736-765 -------
 * For landmark i (state rows id = 3+2i, id+1): sigma_m = the 2x2 block (NOT symmetrised), its
 * eigen-decomposition by Eigen::EigenSolver (:759-761), angle = atan2(V(1,0), V(0,0)) of the FIRST
 * pseudo-eigenvector (:763), x_len = 2 sqrt(5.991 D(0,0)), y_len = 2 sqrt(5.991 D(1,1)) (:764-765).
 * Eigen is not in this image, so the ORDER and SIGN conventions below restate the published
 * algorithm of Eigen 3.3 RealSchur for a 2x2 real matrix (no Hessenberg step; findSmallSubdiagEntry; then
 * splitOffTwoRows = EISPACK hqr2's two-real-roots branch with a Givens rotation): PARITY UNPINNED.
 *   T = [[a, b],[c, d]];  if |c| <= eps (|a|+|d|): already triangular -> D = (a, d), V(:,0) = (1, 0).
 *   else p = (a-d)/2, q = p^2 + c b, z = sqrt|q| (q >= 0 for a covariance block), pz = p >= 0 ? p+z : p-z,
 *        D0 = d + pz, D1 = pz != 0 ? d - c b / pz : D0, V(:,0) = (pz, c)/|(pz, c)|.
 * out: 5 doubles per landmark {mx, my, angle, x_len, y_len}.  Returns the landmark count. */
int oekf_marker_ellipses(const oekf_t *e, double *out5)
{
    const int L = (e->n - 3) / 2;
    const size_t ld = (size_t)e->n;
    for (int i = 0; i < L; ++i) {
        const int id = 3 + 2 * i;
        const double a = e->P[id + id * ld], b = e->P[id + (id + 1) * ld];
        const double c = e->P[(id + 1) + id * ld], d = e->P[(id + 1) + (id + 1) * ld];
        double d0, d1, vx, vy;
        if (fabs(c) <= 2.220446049250313e-16 * (fabs(a) + fabs(d))) {
            d0 = a; d1 = d; vx = 1.0; vy = 0.0;
        } else {
            const double p = 0.5 * (a - d);
            const double q = p * p + c * b;
            const double z = sqrt(fabs(q));
            const double pz = (p >= 0.0) ? p + z : p - z;
            d0 = d + pz;
            d1 = (pz != 0.0) ? d - c * b / pz : d0;
            vx = pz; vy = c;
        }
        out5[5 * i + 0] = e->mu[id];
        out5[5 * i + 1] = e->mu[id + 1];
        out5[5 * i + 2] = atan2(vy, vx);
        out5[5 * i + 3] = 2.0 * sqrt(d0 * 5.991);
        out5[5 * i + 4] = 2.0 * sqrt(d1 * 5.991);
    }
    return L;
}

/* test hook: overwrite the whole filter state (used to start the CPU leg from
 * a state the device path produced). sigma column-major, ld = n. */
void oekf_set_state(oekf_t *e, double time, int n, const double *mu,
                    const double *sigma, const double vt[3])
{
    free(e->mu); free(e->P);
    e->n = n; e->time = time;
    e->mu = (double *)malloc(sizeof(double) * (size_t)n);
    e->P = (double *)malloc(sizeof(double) * (size_t)n * n);
    memcpy(e->mu, mu, sizeof(double) * (size_t)n);
    memcpy(e->P, sigma, sizeof(double) * (size_t)n * n);
    if (vt) memcpy(e->vt, vt, 3 * sizeof(double));
}
void oekf_get_vt(const oekf_t *e, double vt[3]) { memcpy(vt, e->vt, 3 * sizeof(double)); }
void oekf_get_last_match(const oekf_t *e, int *n_state, int *state_pairs,
                         int *n_map, int *map_pairs, int *n_new, int *new_ids)
{
    if (n_state) *n_state = e->n_state;
    if (n_map) *n_map = e->n_map;
    if (n_new) *n_new = e->n_new;
    if (state_pairs) memcpy(state_pairs, e->state_pairs, sizeof(int) * 2 * (size_t)e->n_state);
    if (map_pairs) memcpy(map_pairs, e->map_pairs, sizeof(int) * 2 * (size_t)e->n_map);
    if (new_ids) memcpy(new_ids, e->new_ids, sizeof(int) * (size_t)e->n_new);
}
