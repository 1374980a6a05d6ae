/* scripts/exp_ldexp_check.c -- for EVERY float d: sleef's SSE-form expf as the oracle computes it (oracle/sleef.c, xexpf_v: the scaling by 2^q as
 * ldexpk's five multiplications by powers of two) against the same with the scaling as one correctly rounded ldexpf (what the fused shrink pass
 * uses: devsleef.h xexpf_v_ldexp, v_ldexp_f32).  Prints how many arguments give different bits and the range they lie in, and checks the claim
 * the kernel relies on: they differ only where the result is below FLT_MIN (exp(d) subnormal or flushed by the -104 test), never for a normal result.
 * Arguments whose q = rint(d / ln 2) does not fit an int (|d| >= 1.4885e9; cvtss2si returns the "integer indefinite" there) are skipped: the
 * two forms are not defined to agree there, neither are the host and the device (v_cvt_i32_f32 saturates).
 * build + run: gcc -O2 -fopenmp -ffp-contract=off -msse2 scripts/exp_ldexp_check.c -lm -o /tmp/exp_ldexp_check && /tmp/exp_ldexp_check [stride] */
#include <emmintrin.h>
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define R_LN2f 1.442695040888963407359924681001892137426645954152985934135449406931f
#define L2Uf 0.693145751953125f
#define L2Lf 1.428606765330187045e-06f
static inline int32_t f2i(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
static inline float i2f(int32_t i) { float f; memcpy(&f, &i, 4); return f; }
static inline float mla(float x, float y, float z) { return x * y + z; }
static inline int rint_i(float x) { return _mm_cvt_ss2si(_mm_set_ss(x)); }
static inline float ldexpk_vector(float x, int q)
{
    int m = q >> 31;
    m = (((m + q) >> 6) - m) << 4;
    q = q - (m << 2);
    float u = i2f((int32_t)((uint32_t)(m + 0x7f) << 23));
    x = (((x * u) * u) * u) * u;
    u = i2f((int32_t)((uint32_t)(q + 0x7f) << 23));
    return x * u;
}
static inline float core(float d, int *qq)
{
    int q = rint_i(d * R_LN2f);
    float s = mla((float)q, -L2Uf, d);
    s = mla((float)q, -L2Lf, s);
    float u = 0.00136324646882712841033936f;
    u = mla(u, s, 0.00836596917361021041870117f);
    u = mla(u, s, 0.0416710823774337768554688f);
    u = mla(u, s, 0.166665524244308471679688f);
    u = mla(u, s, 0.499999850988388061523438f);
    u = 1.0f + mla(s * s, u, s);
    *qq = q;
    return u;
}
int main(int argc, char **argv)
{
    const uint64_t stride = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
    uint64_t ndiff = 0, nnormal_diff = 0, ntested = 0, nover = 0;
    float omin = INFINITY;
    float dmin = INFINITY, dmax = -INFINITY, rmax = 0.f;
#pragma omp parallel for schedule(static) reduction(+ : ndiff, nnormal_diff, ntested, nover) reduction(min : dmin, omin) reduction(max : dmax, rmax)
    for (uint64_t b = 0; b < (1ull << 32); b += stride) {
        const float d = i2f((int32_t)(uint32_t)b);
        if (!(fabsf(d) < 1.4885e9f)) continue;          /* NaN, inf, q out of int range */
        int q;
        const float u = core(d, &q);
        float A = ldexpk_vector(u, q), B = ldexpf(u, q);
        if (-104.f > d) A = B = 0.f;
        ++ntested;
        if (f2i(A) != f2i(B) && !(A != A && B != B)) {
            /* results that overflow: ldexpf gives +inf; ldexpk's 2^(16 m) factors leave the exponent field for q >= 512 (d >= 354.6) and the chain
             * returns whatever those bit patterns multiply to.  Counted apart: no caller exponentiates a number that large (and exp of it is not a number) */
            if (B == INFINITY) { ++nover; if (d < omin) omin = d; continue; }
            ++ndiff;
            if (d < dmin) dmin = d;
            if (d > dmax) dmax = d;
            const float big = fabsf(A) > fabsf(B) ? fabsf(A) : fabsf(B);
            if (big > rmax) rmax = big;
            if (big >= FLT_MIN) ++nnormal_diff;
        }
    }
    printf("arguments whose result overflows and where the two forms differ: %llu, the smallest %g\n", (unsigned long long)nover, omin);
    printf("tested %llu arguments (stride %llu): %llu differ, all in d = [%g, %g], largest result among them %g (FLT_MIN = %g); %llu of them with a normal result\n",
           (unsigned long long)ntested, (unsigned long long)stride, (unsigned long long)ndiff, dmin, dmax, rmax, FLT_MIN, (unsigned long long)nnormal_diff);
    return nnormal_diff ? 1 : 0;
}
