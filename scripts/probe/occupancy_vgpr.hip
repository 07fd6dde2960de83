// Probe: residency of 256-thread workgroups (72 KB LDS) as a function of their VGPR / AGPR footprint.  grid = 512 = 2 per CU:
// ~11 us means both were resident, ~22 us means one at a time.
#include <hip/hip_runtime.h>
#include <cstdio>
#define SPIN_KERNEL(NAME, CLOBBER)                                                        \
    __global__ __launch_bounds__(256) void NAME(int *sink, long long ticks)               \
    {                                                                                     \
        extern __shared__ int lds[];                                                      \
        lds[threadIdx.x] = threadIdx.x;                                                   \
        __syncthreads();                                                                  \
        CLOBBER;                                                                          \
        const long long t0 = wall_clock64();                                              \
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);                  \
        if (lds[(threadIdx.x + 1) & 255] == -1) *sink = 1;                                \
    }
SPIN_KERNEL(k_v64, asm volatile("v_mov_b32 v63, 0" ::: "v63"))
SPIN_KERNEL(k_v128, asm volatile("v_mov_b32 v127, 0" ::: "v127"))
SPIN_KERNEL(k_v136a36, asm volatile("v_mov_b32 v135, 0\n v_accvgpr_write_b32 a35, 0" ::: "v135", "a35"))
SPIN_KERNEL(k_v168, asm volatile("v_mov_b32 v167, 0" ::: "v167"))
SPIN_KERNEL(k_v200, asm volatile("v_mov_b32 v199, 0" ::: "v199"))
SPIN_KERNEL(k_v128a128, asm volatile("v_mov_b32 v127, 0\n v_accvgpr_write_b32 a127, 0" ::: "v127", "a127"))
__global__ __launch_bounds__(256) void k_static8(int *sink, long long ticks)        // 8 KB static + dynamic, 172 registers, ~100 SGPRs
{
    extern __shared__ int lds[];
    __shared__ double border[1024];
    border[threadIdx.x] = threadIdx.x;
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    asm volatile("v_mov_b32 v135, 0\n v_accvgpr_write_b32 a35, 0\n s_mov_b32 s99, 0" ::: "v135", "a35", "s99");
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x + 1) & 255] == -1 || border[(threadIdx.x + 3) & 255] == -2.0) *sink = 1;
}
template <class K> static void run(const char *name, K k, int *sink)
{
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    for (int grid : {256, 512}) {
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 72 * 1024, 0, sink, 1000LL);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a, 0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 72 * 1024, 0, sink, 1000LL);
        (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        printf("%-12s grid %4d: %.1f us per launch\n", name, grid, ms * 1e3f / 20);
    }
}
int main()
{
    int *sink; (void)hipMalloc(&sink, 4);
    run("v64", k_v64, sink); run("v128", k_v128, sink); run("v136+a36", k_v136a36, sink);
    {
        (void)hipFuncSetAttribute((const void *)k_static8, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        for (int grid : {256, 512, 545}) {
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            hipLaunchKernelGGL(k_static8, dim3(grid), dim3(256), 64 * 1024, 0, sink, 1000LL);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(a, 0);
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_static8, dim3(grid), dim3(256), 64 * 1024, 0, sink, 1000LL);
            (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
            float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
            printf("static8+dyn64 s99 grid %4d: %.1f us per launch\n", grid, ms * 1e3f / 20);
        }
    }
    run("v168", k_v168, sink); run("v200", k_v200, sink); run("v128+a128", k_v128a128, sink);
    return 0;
}
