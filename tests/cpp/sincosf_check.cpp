// tests/test_detect_cpu.py: the device's restatement of glibc's sinf / cosf (csrc/glibc_sincosf.h), compiled for the HOST and
// checked against the host's libm bit for bit.  Prints "mismatch_sin mismatch_cos samples".
#include <cmath>
#include <cstdint>
#include <cstdio>

#include "glibc_sincosf.h"

int main()
{
    uint64_t s = 88172645463325252ull;
    long bs = 0, bc = 0;
    const long n = 6000000;
    for (long i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        float x = (float)((double)(s >> 11) / 9007199254740992.0 * 40.0 - 20.0);
        if (i % 7 == 0) x *= 1e-3f;
        if (i % 11 == 0) x *= 5.5f;                     // up to |x| = 110
        float a, c;
        if (!glibc_sincosf_core(x, &a, &c)) continue;
        bs += a != sinf(x);
        bc += c != cosf(x);
    }
    printf("%ld %ld %ld\n", bs, bc, n);
    return 0;
}
