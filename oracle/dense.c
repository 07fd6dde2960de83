/*
 * oracle/dense.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Dense column-major FP64 helpers used by the CPU oracle (oracle/ekf_oracle.c):
 * a cache-blocked GEMM and a partial-pivot LU inverse.  They stand in for the
 * two Eigen facilities the reference leans on (dense products at
 * reflector_ekf_slam.cc:178,202,305,306,308,354,355 and MatrixXd::inverse()
 * at :305, which for a dynamic-size matrix is PartialPivLU).  Eigen is not
 * vendored in the reference and is not on this image, so this is a
 * restatement of the published algorithms, not Eigen's code.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this file.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* Threads for the structured-mode baseline's row/column-parallel loops (bench.py's "all cores" figure).  1 = the
 * reference's own build (no OpenMP, CMakeLists.txt:4).  Columns of C are dealt out in chunks that start on multiples of four, so every
 * element sees the same operations in the same order whatever the thread count: results are bit-identical. */
static int od_threads = 1;
void od_set_threads(int n) { od_threads = n < 1 ? 1 : n; }
int od_get_threads(void) { return od_threads; }

static void od_gemm_acc_1(int M, int N, int K, double alpha,
                          const double *A, int lda, const double *B, int ldb,
                          double *C, int ldc);
/* C(MxN) += alpha * A(MxK) * B(KxN), all column-major. */
void od_gemm_acc(int M, int N, int K, double alpha,
                 const double *A, int lda, const double *B, int ldb,
                 double *C, int ldc)
{
    if (od_threads <= 1 || N < 16) { od_gemm_acc_1(M, N, K, alpha, A, lda, B, ldb, C, ldc); return; }
    const int W = 16;                                    /* columns per chunk: a multiple of the kernel's four */
    const int chunks = (N + W - 1) / W;
#pragma omp parallel for schedule(static) num_threads(od_threads)
    for (int c = 0; c < chunks; ++c) {
        const int j0 = c * W, nj = (N - j0 < W) ? (N - j0) : W;
        od_gemm_acc_1(M, nj, K, alpha, A, lda, B + (size_t)j0 * ldb, ldb, C + (size_t)j0 * ldc, ldc);
    }
}
static void od_gemm_acc_1(int M, int N, int K, double alpha,
                          const double *A, int lda, const double *B, int ldb,
                          double *C, int ldc)
{
    enum { MB = 256, KB = 128 };
    for (int k0 = 0; k0 < K; k0 += KB) {
        const int kb = (K - k0 < KB) ? (K - k0) : KB;
        for (int i0 = 0; i0 < M; i0 += MB) {
            const int mb = (M - i0 < MB) ? (M - i0) : MB;
            int j = 0;
            for (; j + 4 <= N; j += 4) {
                double *restrict c0 = C + i0 + (size_t)(j + 0) * ldc;
                double *restrict c1 = C + i0 + (size_t)(j + 1) * ldc;
                double *restrict c2 = C + i0 + (size_t)(j + 2) * ldc;
                double *restrict c3 = C + i0 + (size_t)(j + 3) * ldc;
                int k = 0;
                for (; k + 2 <= kb; k += 2) {
                    const double *restrict a0 = A + i0 + (size_t)(k0 + k) * lda;
                    const double *restrict a1 = a0 + lda;
                    const double b00 = alpha * B[(k0 + k) + (size_t)(j + 0) * ldb];
                    const double b01 = alpha * B[(k0 + k) + (size_t)(j + 1) * ldb];
                    const double b02 = alpha * B[(k0 + k) + (size_t)(j + 2) * ldb];
                    const double b03 = alpha * B[(k0 + k) + (size_t)(j + 3) * ldb];
                    const double b10 = alpha * B[(k0 + k + 1) + (size_t)(j + 0) * ldb];
                    const double b11 = alpha * B[(k0 + k + 1) + (size_t)(j + 1) * ldb];
                    const double b12 = alpha * B[(k0 + k + 1) + (size_t)(j + 2) * ldb];
                    const double b13 = alpha * B[(k0 + k + 1) + (size_t)(j + 3) * ldb];
                    for (int i = 0; i < mb; ++i) {
                        const double x0 = a0[i], x1 = a1[i];
                        c0[i] += x0 * b00 + x1 * b10;
                        c1[i] += x0 * b01 + x1 * b11;
                        c2[i] += x0 * b02 + x1 * b12;
                        c3[i] += x0 * b03 + x1 * b13;
                    }
                }
                for (; k < kb; ++k) {
                    const double *restrict a0 = A + i0 + (size_t)(k0 + k) * lda;
                    const double b00 = alpha * B[(k0 + k) + (size_t)(j + 0) * ldb];
                    const double b01 = alpha * B[(k0 + k) + (size_t)(j + 1) * ldb];
                    const double b02 = alpha * B[(k0 + k) + (size_t)(j + 2) * ldb];
                    const double b03 = alpha * B[(k0 + k) + (size_t)(j + 3) * ldb];
                    for (int i = 0; i < mb; ++i) {
                        const double x0 = a0[i];
                        c0[i] += x0 * b00;
                        c1[i] += x0 * b01;
                        c2[i] += x0 * b02;
                        c3[i] += x0 * b03;
                    }
                }
            }
            for (; j < N; ++j) {
                double *restrict c0 = C + i0 + (size_t)j * ldc;
                for (int k = 0; k < kb; ++k) {
                    const double *restrict a0 = A + i0 + (size_t)(k0 + k) * lda;
                    const double b0 = alpha * B[(k0 + k) + (size_t)j * ldb];
                    for (int i = 0; i < mb; ++i)
                        c0[i] += a0[i] * b0;
                }
            }
        }
    }
}

/* the single-thread kernel on a caller-chosen sub-block (the persistent-team structured update of ekf_oracle.c deals row blocks
 * and 16-column blocks to its threads itself); column blocks must start on multiples of four of the full matrix */
void od_gemm_acc_block(int M, int N, int K, double alpha, const double *A, int lda, const double *B, int ldb, double *C, int ldc)
{
    od_gemm_acc_1(M, N, K, alpha, A, lda, B, ldb, C, ldc);
}

/* C(MxN) = A(MxK) * B(KxN) */
void od_gemm(int M, int N, int K, const double *A, int lda,
             const double *B, int ldb, double *C, int ldc)
{
    for (int j = 0; j < N; ++j)
        memset(C + (size_t)j * ldc, 0, sizeof(double) * (size_t)M);
    od_gemm_acc(M, N, K, 1.0, A, lda, B, ldb, C, ldc);
}

/* B(NxM) = A(MxN)^T */
void od_transpose(int M, int N, const double *A, int lda, double *B, int ldb)
{
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i)
            B[j + (size_t)i * ldb] = A[i + (size_t)j * lda];
}

/*
 * In-place inverse of a dense m x m matrix by LU with partial (row) pivoting
 * followed by solving against the identity: what Eigen's
 * PartialPivLU::inverse() computes for the dynamic-size S at
 * reflector_ekf_slam.cc:305.  Returns 0, or -1 when a pivot is exactly zero.
 */
int od_lu_inverse(int m, double *A, int lda)
{
    int *piv = (int *)malloc(sizeof(int) * (size_t)m);
    double *X = (double *)calloc((size_t)m * m, sizeof(double));
    if (!piv || !X) { free(piv); free(X); return -2; }
    for (int k = 0; k < m; ++k) {
        int p = k;
        double best = fabs(A[k + (size_t)k * lda]);
        for (int i = k + 1; i < m; ++i) {
            const double v = fabs(A[i + (size_t)k * lda]);
            if (v > best) { best = v; p = i; }
        }
        piv[k] = p;
        if (best == 0.0) { free(piv); free(X); return -1; }
        if (p != k)
            for (int j = 0; j < m; ++j) {
                const double t = A[k + (size_t)j * lda];
                A[k + (size_t)j * lda] = A[p + (size_t)j * lda];
                A[p + (size_t)j * lda] = t;
            }
        const double inv = 1.0 / A[k + (size_t)k * lda];
        for (int i = k + 1; i < m; ++i)
            A[i + (size_t)k * lda] *= inv;
        for (int j = k + 1; j < m; ++j) {
            const double akj = A[k + (size_t)j * lda];
            for (int i = k + 1; i < m; ++i)
                A[i + (size_t)j * lda] -= A[i + (size_t)k * lda] * akj;
        }
    }
    /* X = P * I (row-permuted identity), then L y = ., U x = y per column */
    for (int j = 0; j < m; ++j)
        X[j + (size_t)j * m] = 1.0;
    for (int k = 0; k < m; ++k)
        if (piv[k] != k)
            for (int j = 0; j < m; ++j) {
                const double t = X[k + (size_t)j * m];
                X[k + (size_t)j * m] = X[piv[k] + (size_t)j * m];
                X[piv[k] + (size_t)j * m] = t;
            }
    for (int j = 0; j < m; ++j) {
        double *x = X + (size_t)j * m;
        for (int k = 0; k < m; ++k) {
            const double xk = x[k];
            if (xk != 0.0)
                for (int i = k + 1; i < m; ++i)
                    x[i] -= A[i + (size_t)k * lda] * xk;
        }
        for (int k = m - 1; k >= 0; --k) {
            x[k] /= A[k + (size_t)k * lda];
            const double xk = x[k];
            for (int i = 0; i < k; ++i)
                x[i] -= A[i + (size_t)k * lda] * xk;
        }
    }
    for (int j = 0; j < m; ++j)
        memcpy(A + (size_t)j * lda, X + (size_t)j * m, sizeof(double) * (size_t)m);
    free(piv);
    free(X);
    return 0;
}
