// does hipExtAnyOrderLaunch let kernel B start while kernel A (same stream) is still running?  gfx950
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void kA(long long *t, int spin_us)
{
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t0;
    while (wall_clock64() - t0 < 100ll * spin_us) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0) t[1] = wall_clock64();
}
__global__ void kB(long long *t)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) t[2] = wall_clock64();
}
int main()
{
    long long *d, h[3];
    hipMalloc(&d, 64);
    hipStream_t s; hipStreamCreate(&s);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(d, 0, 64);
            hipStreamSynchronize(s);
            hipExtLaunchKernelGGL(kA, dim3(64), dim3(256), 0, s, nullptr, nullptr, 0, d, 50);
            hipExtLaunchKernelGGL(kB, dim3(64), dim3(256), 0, s, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0, d);
            hipStreamSynchronize(s);
            hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            printf("flags %d: A ran %.1f us; B started %.1f us after A's start (%s)\n", mode, (h[1] - h[0]) / 100.0, (h[2] - h[0]) / 100.0,
                   h[2] < h[1] ? "OVERLAPPED" : "after A ended");
        }
    return 0;
}
