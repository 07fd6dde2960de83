// Probe: can the host write straight into fine-grained device memory (large BAR), and what does it cost?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k_sum(const float *p, int n, float *out)
{
    float s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
    atomicAdd(out, s);
}
int main()
{
    const int n = 7200;
    float *d = nullptr, *out = nullptr;
    hipError_t e = hipExtMallocWithFlags((void **)&d, n * 4, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags finegrained: %s ptr %p\n", hipGetErrorString(e), (void *)d);
    if (e != hipSuccess) return 1;
    hipMalloc(&out, 4);
    hipPointerAttribute_t at;
    e = hipPointerGetAttributes(&at, d);
    printf("attr: %s type %d host %p dev %p managed %d\n", hipGetErrorString(e), (int)at.type, at.hostPointer, at.devicePointer, at.isManaged);
    std::vector<float> src(n);
    for (int i = 0; i < n; ++i) src[i] = 1.0f + (i & 7);
    fflush(stdout);
    for (int rep = 0; rep < 5; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        std::memcpy(d, src.data(), n * 4);              // may fault if the BAR does not cover it
        auto t1 = std::chrono::steady_clock::now();
        hipMemset(out, 0, 4);
        hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, 0, d, n, out);
        float r = 0;
        hipMemcpy(&r, out, 4, hipMemcpyDeviceToHost);
        double want = 0; for (int i = 0; i < n; ++i) want += src[i];
        printf("host write %d B: %.2f us; kernel sum %.1f (want %.1f)\n", n * 4, std::chrono::duration<double, std::micro>(t1 - t0).count(), r, want);
        for (int i = 0; i < n; ++i) src[i] += 1.0f;
    }
    return 0;
}
