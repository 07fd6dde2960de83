// per-instruction issue cost on one wave (cycles per instruction, 256 back-to-back copies), gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP256(x) ".rept 256\n" x "\n.endr\n"
#define T(name, body)                                                                                                              \
    __global__ void k_##name(long long *out, double *sink)                                                                         \
    {                                                                                                                               \
        double a = threadIdx.x * 1e-3 + 1.0, b = 1.000001, c = 0.5, d = 0.25;                                                       \
        int i0 = threadIdx.x, i1 = 3;                                                                                               \
        long long best = 1ll << 60;                                                                                                 \
        for (int rep = 0; rep < 4; ++rep) {                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                                      \
            const long long t0 = clock64();                                                                                         \
            asm volatile(REP256(body) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(i0), "+v"(i1)::"vcc", "s20", "s21");               \
            const long long t1 = clock64();                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                                      \
            if (t1 - t0 < best) best = t1 - t0;                                                                                     \
        }                                                                                                                           \
        if (threadIdx.x == 0) out[0] = best;                                                                                        \
        sink[threadIdx.x] = a + b + c + d + i0 + i1;                                                                                \
    }
T(fma_dep, "v_fma_f64 %0, %0, %1, %2")
T(fma_ind, "v_fma_f64 %0, %1, %2, %3\n v_fma_f64 %1, %2, %2, %3")
T(mul_dep, "v_mul_f64 %0, %0, %1")
T(cnd_dep, "v_cndmask_b32 %4, %4, %5, vcc")
T(cnd_ind, "v_cndmask_b32 %4, %5, %5, vcc\n v_cndmask_b32 %5, %5, %5, vcc")
T(mov_dpp, "v_mov_b32_dpp %4, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf")
T(mov64_dpp, "v_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf")
T(fmac_dpp, "v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf")
T(fmac_dpp_self, "v_fmac_f64_dpp %0, %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf")
T(perm16, "v_permlane16_swap_b32 %4, %5")
T(perm32, "v_permlane32_swap_b32 %4, %5")
T(rcp, "v_rcp_f64 %0, %0")
T(readlane, "v_readlane_b32 s20, %4, 5")
T(readlane_use, "v_readlane_b32 s20, %4, 5\n v_mov_b32 %4, s20")
T(mfma_dep, "v_mfma_f64_16x16x4_f64 a[0:7], %0, %1, a[0:7]")
T(mfma_valu, "v_mfma_f64_16x16x4_f64 a[0:7], %0, %1, a[0:7]\n v_fma_f64 %2, %2, %3, %3\n v_fma_f64 %2, %2, %3, %3\n v_fma_f64 %2, %2, %3, %3\n v_fma_f64 %2, %2, %3, %3")
T(mfma_valu32, "v_mfma_f64_16x16x4_f64 a[0:7], %0, %1, a[0:7]\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %4, %4, %5, vcc")
T(mfma4, "v_mfma_f64_4x4x4_4b_f64 %2, %0, %1, %2")
T(mov64, "v_mov_b64 %0, %1")
T(pkmov, "v_pk_mov_b32 %0, %1, %1")
#define RUN(name, n) do { hipLaunchKernelGGL(k_##name, dim3(1), dim3(64), 0, 0, d, s); hipDeviceSynchronize(); long long t; hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost); printf("%-16s %7.2f cycles per instruction (%d per rep)\n", #name, t / (256.0 * n), n); } while (0)
int main()
{
    long long *d; double *s;
    hipMalloc(&d, 64); hipMalloc(&s, 4096);
    RUN(fma_dep, 1); RUN(fma_ind, 2); RUN(mul_dep, 1); RUN(cnd_dep, 1); RUN(cnd_ind, 2); RUN(mov_dpp, 1); RUN(mov64_dpp, 1); RUN(fmac_dpp, 1); RUN(fmac_dpp_self, 1);
    RUN(perm16, 1); RUN(perm32, 1); RUN(rcp, 1); RUN(readlane, 1); RUN(readlane_use, 2); RUN(mfma_dep, 1); RUN(mfma_valu, 5); RUN(mfma_valu32, 5); RUN(mfma4, 1); RUN(mov64, 1); RUN(pkmov, 1);
    return 0;
}
