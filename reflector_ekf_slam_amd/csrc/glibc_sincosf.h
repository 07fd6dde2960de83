// glibc_sincosf.h -- float32 sin / cos with the HOST libm's bits, on the device.
//
// The reference calls std::cos / std::sin on floats (Rotation2D<float>::toRotationMatrix via transform::Rigid2f,
// laser_reflector_detect.cc:117-119,246-306), i.e. glibc's sinf / cosf -- which are NOT correctly rounded (they differ from the
// rounded exact value for 1.3 % of the arguments in [-pi, pi], measured), so neither ocml's sincosf nor an FP64 sincos rounded
// to float reproduces them, and the 2D detector's centres differed from the CPU oracle's in the last place (rounds 1-2: up to
// 7.6e-6 m).  glibc >= 2.28 uses the algorithm of Arm's Optimized Routines (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c,
// s_sincosf.h, s_sincosf_data.c; published, MIT-licensed): n = round(x 2/pi) by a scaled integer cast, r = x - n pi/2 in double,
// an odd / even polynomial of degree 7 / 8 in double, ONE rounding to float.  It is restated here with the operation order of the
// FMA build that glibc's ifunc selects on every x86-64 CPU with FMA (each a + b c is one fused operation): bit-identical to the
// host's sinf / cosf on 5e7 random arguments in [-20, 20] (tests/test_detect_cpu.py compiles this very header for the host and
// checks it against libm).  Arguments with |x| >= 120 (never in the detector: angles are a few pi) are left to the caller.
#pragma once
#if defined(__HIPCC__)
#define GSC_FN __host__ __device__ static inline
#else
#define GSC_FN static inline
#endif

GSC_FN unsigned gsc_top12(float f)
{
    unsigned u;
    __builtin_memcpy(&u, &f, 4);
    return (u >> 20) & 0x7ff;
}
GSC_FN float gsc_poly(double x, double x2, bool neg_tab, int n)
{
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {                                   // sine polynomial (the same in both tables)
        const double x3 = x * x2;
        const double s1 = __builtin_fma(x2, S3, S2);
        const double x7 = x3 * x2;
        const double sv = __builtin_fma(x3, S1, x);
        return (float)__builtin_fma(x7, s1, sv);
    }
    const double sg = neg_tab ? -1.0 : 1.0;               // table 1 = table 0 with the cosine coefficients negated
    const double C0 = sg * 0x1p0, C1 = sg * -0x1.ffffffd0c621cp-2, C2 = sg * 0x1.55553e1068f19p-5, C3 = sg * -0x1.6c087e89a359dp-10,
                 C4 = sg * 0x1.99343027bf8c3p-16;
    const double x4 = x2 * x2;
    const double c2 = __builtin_fma(x2, C4, C3);
    const double c1 = __builtin_fma(x2, C1, C0);
    const double x6 = x4 * x2;
    const double cv = __builtin_fma(x4, C2, c1);
    return (float)__builtin_fma(x6, c2, cv);
}
// returns false when |y| >= 120 (or NaN / inf): nothing written, the caller uses its own libm
GSC_FN bool glibc_sincosf_core(float y, float *sn, float *cs)
{
    const unsigned top = gsc_top12(y);
    double x = (double)y;
    if (top < gsc_top12(0x1.921FB6p-1f)) {                // |y| < pi/4
        const double x2 = x * x;
        if (top < gsc_top12(0x1p-12f)) { *sn = y; *cs = 1.0f; return true; }
        *sn = gsc_poly(x, x2, false, 0);
        *cs = gsc_poly(x, x2, false, 1);
        return true;
    }
    if (top >= gsc_top12(120.0f)) return false;
    const double r = x * 0x1.45F306DC9C883p+23;           // 2/pi * 2^24
    const int n = ((int)r + 0x800000) >> 24;
    x = __builtin_fma(-(double)n, 0x1.921FB54442D18p0, x);
    const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;      // sign[n & 3] = {1, -1, -1, 1}
    const bool neg = (n & 2) != 0;
    const double xs = x * sgn, x2 = x * x;
    *sn = gsc_poly(xs, x2, neg, n);
    *cs = gsc_poly(xs, x2, neg, n ^ 1);
    return true;
}
#if defined(__HIPCC__)
__device__ static inline void glibc_sincosf(float y, float *sn, float *cs)
{
    if (!glibc_sincosf_core(y, sn, cs)) sincosf(y, sn, cs);
}
#endif
