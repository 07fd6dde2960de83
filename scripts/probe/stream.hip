// Probe: what does this MI355X sustain on a P-sized (34.7 MB) FP64 matrix?  write-only, read-only, copy in place (read + write
// every element once = the downdate's algorithmic traffic), with plain 16-byte accesses from many small workgroups.
// Back-to-back launches between one event pair, like bench.py's roofline leg.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_write(v2d *p, size_t n2, double v)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) p[i] = (v2d){v, v};
}
__global__ __launch_bounds__(256) void k_read(const v2d *p, size_t n2, double *sink)
{
    v2d a = {0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) { const v2d x = p[i]; a.x += x.x; a.y += x.y; }
    if (a.x + a.y == 12345.678) *sink = a.x;
}
__global__ __launch_bounds__(256) void k_rmw(v2d *p, size_t n2, double v)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) { v2d x = p[i]; x.x += v; x.y += v; p[i] = x; }
}
template <class F> static float timeit(F f, int reps)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) f();
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) f();
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}
int main()
{
    const size_t ld = 2112, n = 2051, elems = ld * n, n2 = elems / 2;
    v2d *p; double *sink;
    (void)hipMalloc(&p, elems * 8); (void)hipMalloc(&sink, 8);
    (void)hipMemset(p, 0, elems * 8);
    const double MB = elems * 8 / 1e6;
    printf("buffer %.1f MB\n", MB);
    for (int grid : {256, 512, 1024, 2048, 4096}) {
        const float w = timeit([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, p, n2, 1.0); }, 200);
        const float r = timeit([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, p, n2, sink); }, 200);
        const float c = timeit([&] { hipLaunchKernelGGL(k_rmw, dim3(grid), dim3(256), 0, 0, p, n2, 1e-9); }, 200);
        printf("grid %4d: write %.2f us (%.2f TB/s)  read %.2f us (%.2f TB/s)  read+write in place %.2f us (%.2f TB/s of 2x%.1f MB)\n",
               grid, w, MB / w / 1e6 * 1e6 / 1e6, r, MB / r, c, 2 * MB / c, MB);
    }
    const float e = timeit([&] { hipLaunchKernelGGL(k_write, dim3(1), dim3(64), 0, 0, p, (size_t)64, 1.0); }, 200);
    printf("empty-ish launch %.2f us\n", e);
    return 0;
}
