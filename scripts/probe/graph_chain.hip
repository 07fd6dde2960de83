// Probe: does a HIP graph shorten a chain of three dependent, latency-bound kernels (the shape of one EKF update: ~5 us of work
// each, new kernel arguments every iteration)?  (a) three hipLaunchKernelGGL per iteration on one stream; (b) one hipGraphLaunch
// of a three-node graph whose kernel parameters are updated before every launch (hipGraphExecKernelNodeSetParams x 3).
// Reports GPU time per iteration (events around 2000 back-to-back iterations) and host time per iteration.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Args { float obs[64]; int k; };
__global__ __launch_bounds__(256) void k_work(Args a, int *sink, long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (a.obs[threadIdx.x & 63] == -12345.f) *sink = a.k;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    int *sink; CK(hipMalloc(&sink, 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (long long ticks : {100LL, 500LL}) {          // 1 us and 5 us of work per kernel
        Args a = {};
        // (a) plain launches
        for (int i = 0; i < 50; ++i) for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, s, a, sink, ticks);
        CK(hipStreamSynchronize(s));
        auto h0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) { a.k = i; a.obs[i & 63] = (float)i; for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, s, a, sink, ticks); }
        CK(hipEventRecord(e1, s));
        auto h1 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(s));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("work %4.1f us/kernel  launches: GPU %.2f us/iteration, host enqueue %.2f us/iteration\n", ticks * 0.01, ms * 1e3 / iters,
               std::chrono::duration<double, std::micro>(h1 - h0).count() / iters);
        // (b) a three-node graph, parameters updated per launch
        hipGraph_t g; CK(hipGraphCreate(&g, 0));
        hipGraphNode_t node[3];
        void *kargs[3] = {&a, &sink, &ticks};
        hipKernelNodeParams kp = {};
        kp.func = (void *)k_work; kp.gridDim = dim3(64); kp.blockDim = dim3(256); kp.sharedMemBytes = 0; kp.kernelParams = kargs; kp.extra = nullptr;
        for (int k = 0; k < 3; ++k) CK(hipGraphAddKernelNode(&node[k], g, k ? &node[k - 1] : nullptr, k ? 1 : 0, &kp));
        hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        h0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) {
            a.k = i; a.obs[i & 63] = (float)i;
            for (int k = 0; k < 3; ++k) CK(hipGraphExecKernelNodeSetParams(ge, node[k], &kp));
            CK(hipGraphLaunch(ge, s));
        }
        CK(hipEventRecord(e1, s));
        h1 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("work %4.1f us/kernel  graph   : GPU %.2f us/iteration, host enqueue %.2f us/iteration\n", ticks * 0.01, ms * 1e3 / iters,
               std::chrono::duration<double, std::micro>(h1 - h0).count() / iters);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
