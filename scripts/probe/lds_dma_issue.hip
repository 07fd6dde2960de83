// Probe: what does it cost a wave to ISSUE a run of global_load_lds_dwordx4 (the panel DMA of k_downdate2) on gfx950 -- with M0 saved /
// set / restored around every instruction (what dd_dma16 did until round 3), with M0 only set, and with one M0 value per four
// instructions (the instruction's 12-bit offset moves both the global and the LDS address by up to 3 KiB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(256) void k(const double *src, long long *out, size_t ld)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const double *base = src + (size_t)blockIdx.x * 64;
    __syncthreads();
    const long long t0 = clock64();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int pr = 4 * q + wave;
        const double *g = base + (size_t)(2 * (lane & 31)) + (size_t)(2 * pr + (lane >> 5)) * ld;
        const unsigned dst = lds0 + pr * 1024;
        if (MODE == 0) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
        } else if (MODE == 1) {
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(dst) : "memory", "m0");
        } else {
            // wave-major pieces: pr = 8 wave + q; M0 once per four pieces, instruction offset 1024 (q & 3) (global address compensated)
            const int pr2 = 8 * wave + q;
            const char *g2 = (const char *)(base + (size_t)(2 * (lane & 31)) + (size_t)(2 * pr2 + (lane >> 5)) * ld) - 1024 * (q & 3);
            const unsigned dst2 = lds0 + (8 * wave + (q & ~3)) * 1024;
            if ((q & 3) == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(dst2) : "memory", "m0");
            if ((q & 3) == 0) asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(g2) : "memory");
            if ((q & 3) == 1) asm volatile("global_load_lds_dwordx4 %0, off offset:1024" ::"v"(g2) : "memory");
            if ((q & 3) == 2) asm volatile("global_load_lds_dwordx4 %0, off offset:2048" ::"v"(g2) : "memory");
            if ((q & 3) == 3) asm volatile("global_load_lds_dwordx4 %0, off offset:3072" ::"v"(g2) : "memory");
        }
    }
    const long long t1 = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t2 = clock64();
    __syncthreads();
    double s = 0;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) s += smem[i];
    if (threadIdx.x == 0) { out[3 * blockIdx.x] = t1 - t0; out[3 * blockIdx.x + 1] = t2 - t0; out[3 * blockIdx.x + 2] = (long long)s; }
}
template <int MODE> static void run(const char *name, const double *src, long long *out, size_t ld, double expect)
{
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE>), dim3(32), dim3(256), 64 * 64 * 8, 0, src, out, ld);
    hipDeviceSynchronize();
    long long h[96]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    double a = 0, b = 0; bool ok = true;
    for (int i = 0; i < 32; ++i) { a += h[3 * i]; b += h[3 * i + 1]; ok = ok && (double)h[3 * i + 2] == expect; }
    printf("%-58s issue of 8 DMAs: %6.0f cycles, all landed: %6.0f cycles, contents %s\n", name, a / 32, b / 32, ok ? "ok" : "WRONG");
}
int main()
{
    const size_t ld = 2064;
    double *src; long long *out;
    hipMalloc(&src, ld * 64 * 8); hipMalloc(&out, 96 * 8);
    double *h = new double[ld * 64];
    for (size_t i = 0; i < ld * 64; ++i) h[i] = 1.0;
    hipMemcpy(src, h, ld * 64 * 8, hipMemcpyHostToDevice);
    run<0>("M0 saved, set, restored around every instruction", src, out, ld, 4096.0);
    run<1>("M0 set before every instruction", src, out, ld, 4096.0);
    run<2>("M0 set once per four instructions (offset: field)", src, out, ld, 4096.0);
    return 0;
}
