// Probe: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 as k_downdate2 uses it (one wave per SIMD, four independent accumulators,
// operands re-read from LDS every k-step) against the register-only rate.  hipcc --offload-arch=gfx950 -O3 mfma_f64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
template <int MODE, int NACC>
__global__ __launch_bounds__(512) void k(double *out, long long *cyc, int iters)
{
    __shared__ __attribute__((aligned(16))) double lds[64 * 64 * 2];
    for (int i = threadIdx.x; i < 64 * 64 * 2; i += blockDim.x) lds[i] = 1e-3 * (i & 15);
    __syncthreads();
    const int lane = threadIdx.x & 63, idx = lane & 15, kq = lane >> 4;
    const double *aW = lds + 2 * idx + kq * 64, *bK = lds + 4096 + 2 * idx + kq * 64;
    v4d acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    v2d a2 = *(const v2d *)aW, b2 = *(const v2d *)bK;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            v2d a2n = a2, b2n = b2;
            if (MODE == 1) {
                a2n = *(const v2d *)(aW + ((kk + 1) & 15) * 256);
                b2n = *(const v2d *)(bK + ((kk + 1) & 15) * 256);
                __builtin_amdgcn_sched_barrier(0);
            }
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.x, b2.x, acc[0], 0, 0, 0);
            if (NACC > 1) acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.x, b2.y, acc[1], 0, 0, 0);
            if (NACC > 2) acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.y, b2.x, acc[2], 0, 0, 0);
            if (NACC > 3) acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.y, b2.y, acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a2 = a2n; b2 = b2n;
        }
    }
    const long long t1 = clock64();
    double s = 0;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 4; ++r) s += acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE, int NACC> static void run(const char *name, int blocks, int threads, double *out, long long *cyc)
{
    const int iters = 200;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * 16 * NACC;
    const double tf = nm * 2048.0 * (threads / 64) * blocks / (ms * 1e-3) * 1e-12;
    printf("%-44s blocks %4d waves/WG %d: %.1f shader cycles per MFMA per wave, %.1f us, %.1f TFLOP/s\n", name, blocks, threads / 64, (double)c / nm, ms * 1e3, tf);
}
int main()
{
    double *out; long long *cyc;
    hipMalloc(&out, 8 * 512 * 1024); hipMalloc(&cyc, 64);
    run<0, 4>("registers only, 4 accumulators", 256, 256, out, cyc);
    run<0, 4>("registers only, 4 accumulators", 1, 256, out, cyc);
    run<0, 4>("registers only, 4 accumulators", 1, 64, out, cyc);
    run<0, 2>("registers only, 2 accumulators", 256, 256, out, cyc);
    run<0, 1>("registers only, 1 accumulator (dependent)", 256, 256, out, cyc);
    run<1, 4>("operands from LDS every step, 4 accumulators", 256, 256, out, cyc);
    run<1, 4>("operands from LDS every step, 4 accumulators", 256, 512, out, cyc);
    run<1, 2>("operands from LDS every step, 2 accumulators", 256, 512, out, cyc);
    run<1, 4>("operands from LDS every step, 4 accumulators", 1, 64, out, cyc);
    return 0;
}
