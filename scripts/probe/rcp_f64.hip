// How accurate is v_rcp_f64 on gfx950, and what does one / two Newton steps leave?  (k_mid's pivot chain: 1/det)
// hipcc --offload-arch=gfx950 -O3 scripts/probe/rcp_f64.hip -o /tmp/rcp && /tmp/rcp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *r0, double *r1, double *r2, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    double q = __builtin_amdgcn_rcp(d);
    r0[i] = q;
    q = fma(q, fma(-d, q, 1.0), q);
    r1[i] = q;
    q = fma(q, fma(-d, q, 1.0), q);
    r2[i] = q;
}
int main()
{
    const int n = 1 << 22;
    std::vector<double> x(n), a(n), b(n), c(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;          // [0, 1)
        x[i] = ldexp(1.0 + u, (int)(s % 41) - 20);                          // 2^-20 .. 2^21
    }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = 1.0L / (long double)x[i];
        e0 = fmax(e0, (double)fabsl(((long double)a[i] - t) / t));
        e1 = fmax(e1, (double)fabsl(((long double)b[i] - t) / t));
        e2 = fmax(e2, (double)fabsl(((long double)c[i] - t) / t));
    }
    printf("max relative error: v_rcp_f64 %.3e (2^%.1f), + 1 Newton step %.3e (%.2f ulp), + 2 steps %.3e (%.2f ulp)\n", e0, log2(e0), e1,
           e1 / 1.11e-16, e2, e2 / 1.11e-16);
    return 0;
}
