// Probe: the cross-lane exchanges det3d.hip's wave-level sorting networks are built from, checked against lane ^ j:
// quad_perm (1, 2), row_shl / row_shr under bank masks (4, 8), v_permlane16_swap / v_permlane32_swap (16, 32), wave_shr:1.
// hipcc --offload-arch=gfx950 -O3 lane_xor.hip -o lane_xor && ./lane_xor
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int BANK> __device__ inline int dpp(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, BANK, false); }
__device__ inline int lx1(int v) { return dpp<0xB1, 0xf>(v, v); }
__device__ inline int lx2(int v) { return dpp<0x4E, 0xf>(v, v); }
__device__ inline int lx4(int v) { return dpp<0x114, 0xA>(dpp<0x104, 0x5>(v, v), v); }
__device__ inline int lx8(int v) { return dpp<0x118, 0xC>(dpp<0x108, 0x3>(v, v), v); }
__device__ inline int lx16(int v, int lane) { auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false); return (int)((lane & 16) ? r[0] : r[1]); }
__device__ inline int lx32(int v, int lane) { auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false); return (int)((lane & 32) ? r[0] : r[1]); }
__global__ void k(int *o)
{
    const int lane = threadIdx.x;
    const int v = 1000 + lane;
    o[lane] = lx1(v); o[64 + lane] = lx2(v); o[128 + lane] = lx4(v); o[192 + lane] = lx8(v);
    o[256 + lane] = lx16(v, lane); o[320 + lane] = lx32(v, lane);
    o[384 + lane] = dpp<0x138, 0xf>(v, v);
}
int main()
{
    int *d, h[448];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const int js[6] = {1, 2, 4, 8, 16, 32};
    int bad = 0;
    for (int s = 0; s < 6; ++s)
        for (int l = 0; l < 64; ++l)
            if (h[64 * s + l] != 1000 + (l ^ js[s])) { if (bad < 10) printf("xor %d lane %d: got %d\n", js[s], l, h[64 * s + l] - 1000); ++bad; }
    for (int l = 1; l < 64; ++l) if (h[384 + l] != 1000 + l - 1) { if (bad < 20) printf("wave_shr lane %d: got %d\n", l, h[384 + l] - 1000); ++bad; }
    printf("lane 0 of wave_shr:1 = %d (old kept = 1000)\n", h[384]);
    printf(bad ? "MISMATCHES: %d\n" : "all exchanges = lane ^ j (%d mismatches)\n", bad);
    return bad != 0;
}
