// How accurate is v_rcp_f64 by itself, and after one / two Newton steps (leaf_inverse16 uses two)?  hipcc --offload-arch=gfx950 rcp_f64.hip -o rcp_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <random>
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
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-12.0, 12.0), m(1.0, 2.0);
    for (int i = 0; i < n; ++i) x[i] = std::ldexp(m(g), (int)u(g)) * ((i & 1) ? 1 : 1);
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, 8 * n); hipMalloc(&d0, 8 * n); hipMalloc(&d1, 8 * n); hipMalloc(&d2, 8 * n);
    hipMemcpy(dx, x.data(), 8 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, 8 * n, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0; long ne1 = 0, ne2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = 1.0L / (long double)x[i];
        const double ex = (double)t;
        e0 = std::fmax(e0, std::fabs((double)(((long double)a[i] - t) / t)));
        e1 = std::fmax(e1, std::fabs((double)(((long double)b[i] - t) / t)));
        e2 = std::fmax(e2, std::fabs((double)(((long double)c[i] - t) / t)));
        ne1 += b[i] != ex; ne2 += c[i] != ex;
    }
    std::printf("max relative error: v_rcp_f64 %.3e   + one Newton step %.3e (%ld of %d not the correctly rounded 1/x)   + two %.3e (%ld)\n", e0, e1, ne1, n, e2, ne2);
    return 0;
}
