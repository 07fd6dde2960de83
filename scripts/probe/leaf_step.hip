// What does ONE 2x2 pivot step of k_mid's in-wave 16x16 Gauss-Jordan leaf cost, alone on a CU?  One workgroup of 512 threads
// (as in k_mid), wave 0 runs the leaf on an SPD block and stamps s_memtime after every step; variants strip parts of the step.
// hipcc --offload-arch=gfx950 -O3 scripts/probe/leaf_step.hip -o scripts/probe/leaf_step.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

template <int VAR>
__global__ __launch_bounds__(512) void k(const double *A, double *out, long long *ticks)
{
    __shared__ __attribute__((aligned(16))) double lp[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave != 0) { __syncthreads(); return; }
    const int g = lane >> 4, c = lane & 15;
    v4d a;
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = A[(g + 4 * r) * 16 + c];
    double *rowbuf = lp, *colbuf = lp + 32;
    long long tk[9];
    tk[0] = clock64();
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int k = 2 * kk, rk = k >> 2, gk = k & 3;
        if (VAR != 2) {
            if (g == gk || g == gk + 1) rowbuf[2 * c + (g - gk)] = a[rk];
            if (c == k || c == k + 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) colbuf[(4 * g + r) * 2 + (c - k)] = a[r];
            }
        }
        __builtin_amdgcn_wave_barrier();
        v2d dA, dB, xx, ff[4];
        if (VAR != 2) {
            dA = *(const v2d *)(rowbuf + 2 * k); dB = *(const v2d *)(rowbuf + 2 * k + 2);
            xx = *(const v2d *)(rowbuf + 2 * c);
#pragma unroll
            for (int r = 0; r < 4; ++r) ff[r] = *(const v2d *)(colbuf + (4 * g + r) * 2);
        } else {                                            // VAR 2: no LDS at all (operands from registers: wrong numbers, timing only)
            dA = (v2d){a[0] + 2.0, a[1] * 1e-3}; dB = (v2d){a[2] * 1e-3, a[3] + 2.0};
            xx = (v2d){a[rk], a[(rk + 1) & 3]};
#pragma unroll
            for (int r = 0; r < 4; ++r) ff[r] = (v2d){a[r] * 0.5, a[(r + 1) & 3] * 0.25};
        }
        __builtin_amdgcn_wave_barrier();
        const double d00 = dA.x, d10 = dA.y, d01 = dB.x, d11 = dB.y;
        const double det = fma(d00, d11, -(d01 * d10));
        const bool pc0 = c == k, pc1 = c == k + 1;
        double x0 = xx.x, x1 = xx.y;
        if (VAR != 3) {
            if (pc0) { x0 = 1.0; x1 = 0.0; }
            if (pc1) { x0 = 0.0; x1 = 1.0; }
        }
        const double t0 = fma(d11, x0, -(d01 * x1)), t1 = fma(d00, x1, -(d10 * x0));
        double u[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) u[r] = fma(ff[r].y, t1, ff[r].x * t0);
        double q;
        if (VAR == 1) q = det;
        else {
            const double q0 = __builtin_amdgcn_rcp(det);
            const double e0 = fma(-det, q0, 1.0);
            const double q1 = fma(q0, e0, q0), e1 = e0 * e0;
            q = fma(q1, e1, q1);
        }
        const double R0 = t0 * q, R1 = t1 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double base = (VAR != 3 && (pc0 || pc1)) ? 0.0 : a[r];
            double v = fma(-u[r], q, base);
            if (VAR != 3 && r == rk) v = (g == gk) ? R0 : ((g == gk + 1) ? R1 : v);
            a[r] = v;
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 0" ::"v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));      // the stamp waits for the step's results
        tk[kk + 1] = clock64();
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(g + 4 * r) * 16 + c] = a[r];
    if (lane == 0) for (int i = 0; i < 9; ++i) ticks[i] = tk[i];
    __syncthreads();
}

template <int VAR> static void run(const double *dA, double *dO, long long *dT, const char *name)
{
    long long t[9];
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k<VAR>, dim3(1), dim3(512), 0, 0, dA, dO, dT);
        hipDeviceSynchronize();
        hipMemcpy(t, dT, sizeof(t), hipMemcpyDeviceToHost);
        printf("%-34s rep %d: cycles per step:", name, rep);
        for (int i = 0; i < 8; ++i) printf(" %lld", t[i + 1] - t[i]);
        printf("  total %lld\n", t[8] - t[0]);
    }
}
int main()
{
    std::vector<double> A(256);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) A[i * 16 + j] = (i == j ? 4.0 : 0.0) + 1.0 / (1 + i + j);
    double *dA, *dO; long long *dT;
    hipMalloc(&dA, 2048); hipMalloc(&dO, 2048); hipMalloc(&dT, 128);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
    run<0>(dA, dO, dT, "full step");
    std::vector<double> O(256);
    hipMemcpy(O.data(), dO, 2048, hipMemcpyDeviceToHost);
    double err = 0;                                         // check: A * inv = I
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0;
            for (int l = 0; l < 16; ++l) s += A[i * 16 + l] * O[l * 16 + j];
            err = fmax(err, fabs(s - (i == j)));
        }
    printf("max |A inv(A) - I| = %.2e\n", err);
    run<1>(dA, dO, dT, "no reciprocal chain");
    run<2>(dA, dO, dT, "no LDS exchange");
    run<3>(dA, dO, dT, "no selects (unit vectors, R rows)");
    return 0;
}
