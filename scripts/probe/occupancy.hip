// Probe: how many 256-thread workgroups with a given LDS footprint does one MI355X CU hold at once?
// Every workgroup spins for ~10 us; grid = 512 = 2 per CU.  One round (~10 us) means 2 resident per CU, two rounds ~20 us.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_spin(int *sink, long long ticks)
{
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x + 1) & 255] == -1) *sink = 1;
}
int main()
{
    int *sink; (void)hipMalloc(&sink, 4);
    for (int kb : {8, 40, 52, 64, 72, 80, 96, 128}) {
        (void)hipFuncSetAttribute((const void *)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
        for (int grid : {256, 512, 768, 1024}) {
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            hipLaunchKernelGGL(k_spin, dim3(grid), dim3(256), kb * 1024, 0, sink, 1000LL);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(a, 0);
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_spin, dim3(grid), dim3(256), kb * 1024, 0, sink, 1000LL);   // 1000 ticks of 100 MHz = 10 us
            (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
            float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
            printf("LDS %3d KB grid %4d: %.1f us per launch\n", kb, grid, ms * 1e3f / 20);
        }
    }
    return 0;
}
