// Probe for k_mid's leaf: 16x16 SPD inverse in one wave by the SYMMETRIC SWEEP with SCALAR pivots, every step ONE v_mfma_f64_16x16x4 and
// no LDS exchange at all.  C layout: lane (g, c) = (lane >> 4, lane & 15), register r <-> element (g + 4 r, c).  Pivot k = row k = register
// k / 4 of lane group k % 4: that register IS the pivot row as a B operand (k-slice = the group), and -- the matrix being symmetric -- the
// pivot column as an A operand (A_op[m][slice] sits in lane (slice, m), which holds A(k, m) = A(m, k)).  One step, q = 1 / A(k,k):
//     A_op[m] = (m == k) ? q : -A(k,m) q   (group k % 4 only, 0 elsewhere),   B_op[n] = (n == k) ? -1 : A(k,n),   C_in = A with row k and column k zeroed
//     A <- C_in + A_op B_op :   A(i,j) - A(i,k) A(k,j) q  |  row k: A(k,j) q  |  column k: A(i,k) q  |  (k,k): -q          (Goodnight's sweep)
// After 16 steps A = -A0^-1.  Variant 1 takes the NEXT pivot from the operands before the MFMA that produces it (three v_readlane pairs, one
// FMA) so that the reciprocal chain runs under the MFMA's latency.
// hipcc --offload-arch=gfx950 -O3 scripts/probe/leaf_sweep1.hip -o scripts/probe/leaf_sweep1.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));

__device__ static inline double rcp_nr(double d)
{
    double q = __builtin_amdgcn_rcp(d);
    q = fma(q, fma(-d, q, 1.0), q);
    return fma(q, fma(-d, q, 1.0), q);
}
__device__ static inline double lane_bcast(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

template <int VARIANT>
__device__ static inline bool leaf_sweep1(v4d &a, int g, int c)
{
    bool bad = false;
    double q = 0;
    if (VARIANT == 1) { const double p = lane_bcast(a[0], 0); bad |= !(p > 0.0); q = rcp_nr(p); }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r0 = k >> 2, g0 = k & 3;
        const bool isg = g == g0, isc = c == k;
        if (VARIANT == 0) { const double p = lane_bcast(a[r0], 16 * g0 + k); bad |= !(p > 0.0); q = rcp_nr(p); }
        const double v = a[r0];
        const double w = isc ? q : -(v * q);
        const double aop = isg ? w : 0.0;
        const double bop = isc ? -1.0 : v;
        v4d cin = a;
#pragma unroll
        for (int r = 0; r < 4; ++r) cin[r] = isc ? 0.0 : cin[r];
        cin[r0] = isg ? 0.0 : cin[r0];
        a = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, cin, 0, 0, 0);
        if (VARIANT == 1 && k < 15) {
            // the next pivot A(k+1,k+1) = C_in(k+1,k+1) + A_op[k+1] B_op[k+1], from what went INTO the MFMA
            const int k1 = k + 1, r1 = k1 >> 2, g1 = k1 & 3;
            const double pa = lane_bcast(aop, 16 * g0 + k1), pb = lane_bcast(bop, 16 * g0 + k1), pc = lane_bcast(cin[r1], 16 * g1 + k1);
            const double p = fma(pa, pb, pc);
            bad |= !(p > 0.0);
            q = rcp_nr(p);
        }
    }
    a = -a;
    return bad;
}


// zero as far as FP64 arithmetic can tell: the high dword cleared (what is left is a positive denormal below 2^-1042)
__device__ static inline double zero_hi(double v, bool z) { return __hiloint2double(z ? 0 : __double2hiint(v), __double2loint(v)); }

// Variant 2: the next pivot by symmetry from row k+1 of the matrix BEFORE this step (A(k+1,k+1) - A(k+1,k)^2 q: two v_readlane pairs of one
// register, issued first thing), its reciprocal chain issued right behind the MFMA; zeroing touches the high dword only.
__device__ static inline bool leaf_sweep1_v2(v4d &a, int g, int c)
{
    bool bad = false;
    double q;
    { const double p = lane_bcast(a[0], 0); bad |= !(p > 0.0); q = rcp_nr(p); }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r0 = k >> 2, g0 = k & 3, k1 = (k + 1) & 15, r1 = k1 >> 2, g1 = k1 & 3;
        const bool isg = g == g0, isc = c == k;
        const double x = lane_bcast(a[r1], 16 * g1 + k), d = lane_bcast(a[r1], 16 * g1 + k1);
        const double v = a[r0];
        const double w = isc ? q : -(v * q);
        const double aop = zero_hi(w, !isg);
        const double bop = isc ? -1.0 : v;
        v8i ai = __builtin_bit_cast(v8i, a);
#pragma unroll
        for (int r = 0; r < 4; ++r) ai[2 * r + 1] = (r == r0 ? (isc || isg) : isc) ? 0 : ai[2 * r + 1];
        a = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, __builtin_bit_cast(v4d, ai), 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (k < 15) {
            const double p = fma(-(x * q), x, d);
            bad |= !(p > 0.0);
            q = rcp_nr(p);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    a = -a;
    return bad;
}

// Variant 3: variant 0 (pivot read back with v_readlane, reciprocal on the chain) with the zeroing on the high dwords only, in place.
__device__ static inline bool leaf_sweep1_v3(v4d &a, int g, int c)
{
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r0 = k >> 2, g0 = k & 3;
        const bool isg = g == g0, isc = c == k;
        const double p = lane_bcast(a[r0], 16 * g0 + k);
        bad |= !(p > 0.0);
        const double q = rcp_nr(p);
        const double v = a[r0];
        const double w = isc ? q : -(v * q);
        v8i ai = __builtin_bit_cast(v8i, a);
        const double aop = __hiloint2double(isg ? __double2hiint(w) : 0, __double2loint(w));
        const double bop = isc ? -1.0 : v;
#pragma unroll
        for (int r = 0; r < 4; ++r) ai[2 * r + 1] = (r == r0 ? (isc || isg) : isc) ? 0 : ai[2 * r + 1];
        a = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, __builtin_bit_cast(v4d, ai), 0, 0, 0);
    }
    a = -a;
    return bad;
}

// Variant 4: variant 3 with the instruction sequence written out (the compiler copies the eight-register tuple around every step):
// the matrix pinned to v[40:47], one asm statement per pivot, the MFMA's result hazard (19 wait states) closed inside the statement.
//   s[20:21] pivot   s[22:23] lanes c == k   s[24:25] lanes of group k % 4   s[26:27] their union   v[48:49] q   v[50:51] Newton residual
//   v[52:53] A operand   v[54:55] B operand   v[56:57] -(v q)   %[m1hi] = high dword of -1.0
template <int K> __device__ __forceinline__ void sweep_step_asm(v4d &a, int &minhi, int m1hi)
{
    constexpr int r0 = K >> 2, g0 = K & 3;
    constexpr unsigned mc = 0x00010001u << K, mglo = g0 < 2 ? 0xffffu << (16 * g0) : 0u, mghi = g0 < 2 ? 0u : 0xffffu << (16 * (g0 - 2));
    asm volatile(
        "v_readlane_b32 s20, v[40+2*%c[r0]], %c[lane]\n"
        "v_readlane_b32 s21, v[41+2*%c[r0]], %c[lane]\n"
        "s_mov_b32 s22, %c[mc]\n"
        "s_mov_b32 s23, %c[mc]\n"
        "v_rcp_f64 v[48:49], s[20:21]\n"
        "s_mov_b32 s24, %c[mglo]\n"
        "s_mov_b32 s25, %c[mghi]\n"
        "s_or_b64 s[26:27], s[22:23], s[24:25]\n"
        "s_min_i32 %[minhi], %[minhi], s21\n"
        "v_fma_f64 v[50:51], -s[20:21], v[48:49], 1.0\n"
        "v_fma_f64 v[48:49], v[48:49], v[50:51], v[48:49]\n"
        "v_fma_f64 v[50:51], -s[20:21], v[48:49], 1.0\n"
        "v_fma_f64 v[48:49], v[48:49], v[50:51], v[48:49]\n"
        "v_mul_f64 v[56:57], v[40+2*%c[r0]:41+2*%c[r0]], -v[48:49]\n"
        "v_cndmask_b32 v54, v[40+2*%c[r0]], 0, s[22:23]\n"
        "v_cndmask_b32 v55, v[41+2*%c[r0]], %[m1hi], s[22:23]\n"
        "v_cndmask_b32 v41, v41, 0, s[%c[z0]:%c[z0]+1]\n"
        "v_cndmask_b32 v43, v43, 0, s[%c[z1]:%c[z1]+1]\n"
        "v_cndmask_b32 v45, v45, 0, s[%c[z2]:%c[z2]+1]\n"
        "v_cndmask_b32 v47, v47, 0, s[%c[z3]:%c[z3]+1]\n"
        "v_cndmask_b32 v52, v56, v48, s[22:23]\n"
        "v_cndmask_b32 v53, v57, v49, s[22:23]\n"
        "v_cndmask_b32 v53, 0, v53, s[24:25]\n"
        "s_nop 1\n"
        "v_mfma_f64_16x16x4_f64 v[40:47], v[52:53], v[54:55], v[40:47]\n"
        "s_nop 15\n"
        "s_nop 2\n"
        : "+{v[40:47]}"(a), [minhi] "+s"(minhi)
        : [r0] "n"(r0), [lane] "n"(16 * g0 + K), [mc] "n"(mc), [mglo] "n"(mglo), [mghi] "n"(mghi), [m1hi] "v"(m1hi),
          [z0] "n"(r0 == 0 ? 26 : 22), [z1] "n"(r0 == 1 ? 26 : 22), [z2] "n"(r0 == 2 ? 26 : 22), [z3] "n"(r0 == 3 ? 26 : 22)
        : "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
}
template <int K> struct SweepSteps {
    template <class Side> static __device__ __forceinline__ void run(v4d &a, int &minhi, int m1hi, Side &&side)
    {
        SweepSteps<K - 1>::run(a, minhi, m1hi, side);
        sweep_step_asm<K>(a, minhi, m1hi);
        if (K & 1) side(K >> 1);
    }
};
template <> struct SweepSteps<-1> { template <class Side> static __device__ __forceinline__ void run(v4d &, int &, int, Side &&) {} };
__device__ static inline bool leaf_sweep1_v4(v4d &a, int g, int c)
{
    (void)g; (void)c;
    int minhi = 0x7fffffff;
    SweepSteps<15>::run(a, minhi, (int)0xbff00000, [](int) {});
    a = -a;
    return minhi <= 0 || !(a[0] == a[0]);                 // a pivot <= 0, or a NaN (it spreads over the whole block)
}

template <int VARIANT>
__global__ __launch_bounds__(512) void k(const double *A, double *out, long long *ticks, int nrep)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave != 0) return;
    const int g = lane >> 4, c = lane & 15;
    v4d a;
    long long best = 1ll << 60, first = 0;
    bool bad = false;
    for (int rep = 0; rep < nrep; ++rep) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = __builtin_nontemporal_load(&A[(g + 4 * r) * 16 + c]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const long long t0 = clock64();
        __builtin_amdgcn_sched_barrier(0);
        bad |= (VARIANT == 4) ? leaf_sweep1_v4(a, g, c) : (VARIANT == 3) ? leaf_sweep1_v3(a, g, c) : (VARIANT == 2) ? leaf_sweep1_v2(a, g, c) : leaf_sweep1<VARIANT >= 2 ? 0 : VARIANT>(a, g, c);
        asm volatile("s_nop 0" ::"v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
        __builtin_amdgcn_sched_barrier(0);
        const long long t1 = clock64();
        if (rep == 0) first = t1 - t0;
        else if (t1 - t0 < best) best = t1 - t0;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(g + 4 * r) * 16 + c] = a[r];
    if (lane == 0) { ticks[0] = best; ticks[1] = bad; ticks[2] = first; }
}

template <int VARIANT> static void run(const std::vector<double> &A, double *dA, double *dO, long long *dT)
{
    std::vector<double> O(256);
    hipLaunchKernelGGL(k<VARIANT>, dim3(1), dim3(512), 0, 0, dA, dO, dT, 8);
    hipDeviceSynchronize();
    long long t[3];
    hipMemcpy(t, dT, sizeof(t), hipMemcpyDeviceToHost);
    hipMemcpy(O.data(), dO, 2048, hipMemcpyDeviceToHost);
    double err = 0, asym = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0;
            for (int l = 0; l < 16; ++l) s += A[i * 16 + l] * O[l * 16 + j];
            err = fmax(err, fabs(s - (i == j)));
            asym = fmax(asym, fabs(O[i * 16 + j] - O[j * 16 + i]));
        }
    printf("variant %d: leaf %lld cycles warm (%.0f per pivot; first pass %lld), bad %lld, max |A inv - I| = %.2e, asymmetry %.2e\n", VARIANT, t[0],
           t[0] / 16.0, t[2], t[1], err, asym);
}

int main()
{
    std::vector<double> A(256);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) A[i * 16 + j] = (i == j ? 4.0 + 0.37 * i : 0.0) + 1.0 / (1 + i + j);
    double *dA, *dO; long long *dT;
    hipMalloc(&dA, 2048); hipMalloc(&dO, 2048); hipMalloc(&dT, 128);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) { run<0>(A, dA, dO, dT); run<1>(A, dA, dO, dT); run<2>(A, dA, dO, dT); run<3>(A, dA, dO, dT); run<4>(A, dA, dO, dT); }
    return 0;
}
