// Probe for k_mid's leaf: 16x16 SPD inverse in one wave by the SYMMETRIC SWEEP with 4x4 pivots, the rank-4 updates on the matrix pipe.
// C layout: lane (g, c) = (lane >> 4, lane & 15), register r <-> element (g + 4 r, c).  Pivot set K = rows 4 kk .. 4 kk + 3 = register kk of
// the four lane groups: a[kk] IS the 4 x 16 pivot-row panel in MFMA B-operand layout, and -- the matrix being symmetric -- also the A operand
// (A_op[i][k] = A(k, i)).  One step:  D = A(K,K) through LDS to every lane; D^-1 (4x4, in-lane, 2x2 block formulas); R = D^-1 A(K,:) by ONE
// MFMA (D^-1 as a zero-padded A operand); R' = R with its K columns replaced by I - D^-1; A -= A(K,:)^T R' by ONE MFMA (updates all 256
// entries, leaves [I | 0] in the pivot rows); pivot rows := [-D^-1 | R].  After four steps A = -A0^-1.
// hipcc --offload-arch=gfx950 -O3 scripts/probe/leaf_sweep4.hip -o scripts/probe/leaf_sweep4.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

__device__ static inline double rcp_nr(double d)
{
    double q = __builtin_amdgcn_rcp(d);
    const double e0 = fma(-d, q, 1.0);
    q = fma(q, e0, q);
    return fma(q, e0 * e0, q);
}

// in place: a <- inverse (not negated on return)
__device__ static inline bool leaf_sweep16(v4d &a, int g, int c, double *lp)
{
    bool bad = false;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int c4 = c - 4 * kk;                                  // column index inside the pivot block, if 0 <= c4 < 4
        const bool incol = c4 >= 0 && c4 < 4;
        if (incol) lp[4 * g + c4] = a[kk];                          // D(g, c4)
        __builtin_amdgcn_wave_barrier();
        const v2d r0 = *(const v2d *)(lp + 0), r0b = *(const v2d *)(lp + 2);
        const v2d r1 = *(const v2d *)(lp + 4), r1b = *(const v2d *)(lp + 6);
        const v2d r2 = *(const v2d *)(lp + 8), r2b = *(const v2d *)(lp + 10);
        const v2d r3 = *(const v2d *)(lp + 12), r3b = *(const v2d *)(lp + 14);
        __builtin_amdgcn_wave_barrier();
        // lower triangle of D
        const double d00 = r0.x, d10 = r1.x, d11 = r1.y, d20 = r2.x, d21 = r2.y, d22 = r2b.x, d30 = r3.x, d31 = r3.y, d32 = r3b.x, d33 = r3b.y;
        (void)r0b; (void)r1b;
        // A11^-1
        const double det1 = fma(d00, d11, -(d10 * d10));
        const double q1 = rcp_nr(det1);
        const double i00 = d11 * q1, i10 = -d10 * q1, i11 = d00 * q1;
        // T = A21 A11^-1  (A21 = [[d20, d21], [d30, d31]])
        const double t00 = fma(d20, i00, d21 * i10), t01 = fma(d20, i10, d21 * i11);
        const double t10 = fma(d30, i00, d31 * i10), t11 = fma(d30, i10, d31 * i11);
        // S = A22 - T A21^T (symmetric)
        const double s00 = d22 - fma(t00, d20, t01 * d21), s10 = d32 - fma(t10, d20, t11 * d21), s11 = d33 - fma(t10, d30, t11 * d31);
        const double det2 = fma(s00, s11, -(s10 * s10));
        if (!(det1 > 0.0) || !(d00 > 0.0) || !(det2 > 0.0) || !(s00 > 0.0)) bad = true;
        const double q2 = rcp_nr(det2);
        const double j00 = s11 * q2, j10 = -s10 * q2, j11 = s00 * q2;          // B22 = S^-1
        // B21 = -J T
        const double b00 = -fma(j00, t00, j10 * t10), b01 = -fma(j00, t01, j10 * t11);
        const double b10 = -fma(j10, t00, j11 * t10), b11 = -fma(j10, t01, j11 * t11);
        // B11 = A11^-1 - T^T B21 (symmetric)
        const double e00 = i00 - fma(t00, b00, t10 * b10), e10 = i10 - fma(t01, b00, t11 * b10), e11 = i11 - fma(t01, b01, t11 * b11);
        // full symmetric inverse Dinv(x, y): rows x = 0..3
        //   [ e00 e10 b00 b10 ]
        //   [ e10 e11 b01 b11 ]
        //   [ b00 b01 j00 j10 ]
        //   [ b10 b11 j10 j11 ]
        // this lane's entry Dinv(x, g), x = c (A operand of the R product, c < 4) or c4 (the K columns of R')
        const int x = incol ? c4 : (c & 3);
        const double col0 = (x == 0) ? e00 : (x == 1) ? e10 : (x == 2) ? b00 : b10;     // Dinv(x, 0)
        const double col1 = (x == 0) ? e10 : (x == 1) ? e11 : (x == 2) ? b01 : b11;     // Dinv(x, 1)
        const double col2 = (x == 0) ? b00 : (x == 1) ? b01 : (x == 2) ? j00 : j10;     // Dinv(x, 2)
        const double col3 = (x == 0) ? b10 : (x == 1) ? b11 : (x == 2) ? j10 : j11;     // Dinv(x, 3)
        const double dxg = (g == 0) ? col0 : (g == 1) ? col1 : (g == 2) ? col2 : col3;   // Dinv(x, g) = Dinv(g, x)
        // R = D^-1 A(K, :): A operand lane (g, c) = Dinv(c, g) for c < 4, else 0
        const double aop = (c < 4) ? ((incol && kk != 0) ? 0.0 : dxg) : 0.0;
        // (for kk = 0 the two uses coincide: x = c4 = c; for kk > 0 lanes c < 4 are not pivot columns: x = c & 3 = c)
        const v4d zero = {0, 0, 0, 0};
        const v4d Rv = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, a[kk], zero, 0, 0, 0);
        const double R = Rv[0];                                    // R(g, c)
        // R' : K columns replaced by I - D^-1
        const double Rp = incol ? ((g == c4 ? 1.0 : 0.0) - dxg) : R;
        // A -= A(K,:)^T R'
        a = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[kk], Rp, a, 0, 0, 0);
        // pivot rows := [-D^-1 | R]
        a[kk] = incol ? -dxg : R;
    }
    a = -a;
    return bad;
}

__global__ __launch_bounds__(512) void k(const double *A, double *out, long long *ticks)
{
    __shared__ __attribute__((aligned(16))) double lp[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave != 0) { __syncthreads(); return; }
    const int g = lane >> 4, c = lane & 15;
    v4d a;
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = A[(g + 4 * r) * 16 + c];
    const long long t0 = clock64();
    const bool bad = leaf_sweep16(a, g, c, lp);
    asm volatile("s_nop 0" ::"v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
    const long long t1 = clock64();
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(g + 4 * r) * 16 + c] = a[r];
    if (lane == 0) { ticks[0] = t1 - t0; ticks[1] = bad; }
    __syncthreads();
}

int main()
{
    std::vector<double> A(256), O(256);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) A[i * 16 + j] = (i == j ? 4.0 : 0.0) + 1.0 / (1 + i + j);
    double *dA, *dO; long long *dT;
    hipMalloc(&dA, 2048); hipMalloc(&dO, 2048); hipMalloc(&dT, 128);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, dA, dO, dT);
        hipDeviceSynchronize();
        long long t[2];
        hipMemcpy(t, dT, sizeof(t), hipMemcpyDeviceToHost);
        hipMemcpy(O.data(), dO, 2048, hipMemcpyDeviceToHost);
        double err = 0, asym = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double s = 0;
                for (int l = 0; l < 16; ++l) s += A[i * 16 + l] * O[l * 16 + j];
                err = fmax(err, fabs(s - (i == j)));
                asym = fmax(asym, fabs(O[i * 16 + j] - O[j * 16 + i]));
            }
        printf("rep %d: leaf %lld cycles (%.0f per 4 rows), bad %lld, max |A inv - I| = %.2e, asymmetry %.2e\n", rep, t[0], t[0] / 4.0, t[1], err, asym);
    }
    return 0;
}
