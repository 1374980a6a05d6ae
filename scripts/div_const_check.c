/* scripts/div_const_check.c -- division by a compile-time constant as one multiplication and two fused multiply-adds:
 *     R = RN(1 / C);   q = x * R;   r = fma(-C, q, x);   q' = fma(r, R, q)
 * (Markstein's correction step with the reciprocal known in advance: three full-rate instructions where the correctly rounded x / C of the
 * compiler's expansion costs seventeen issue slots on gfx950 -- div_scale x 2, rcp, six fma / mul, div_fmas, div_fixup).  Whether q' is the
 * correctly rounded quotient for EVERY x depends on C, so it is not argued but walked: for every float x (all 2^32 bit patterns) q' is compared
 * with x / C bit for bit, NaN against NaN.  Prints, per constant, how many x differ and the range of |x| they lie in -- the device code
 * (devmath.h div_const) takes the short form only for |x| inside [lo, hi] and the compiler's division outside.
 * build + run: gcc -O2 -fopenmp -ffp-contract=off -mfma scripts/div_const_check.c -lm -o /tmp/div_const_check && /tmp/div_const_check [stride] */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
int main(int argc, char **argv)
{
    const uint64_t stride = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
    /* the guard the device code uses: the short form for 2^-100 <= |x| <= 2^100 or x == 0 */
    const float glo = 0x1p-100f, ghi = 0x1p100f;
    const float consts[] = {65535.f, 3.f, 0.9642f, 0.8249f, 60.f};
    int bad = 0;
    for (unsigned ci = 0; ci < sizeof consts / sizeof consts[0]; ++ci) {
        volatile float Cv = consts[ci];
        const float C = Cv, R = 1.0f / C;
        uint64_t ndiff = 0, ndiff_guarded = 0, tested = 0;
        float dmin = INFINITY, dmax = 0.f;
#pragma omp parallel for reduction(+ : ndiff, ndiff_guarded, tested) reduction(min : dmin) reduction(max : dmax) schedule(static)
        for (uint64_t i = 0; i < (1ull << 32); i += stride) {
            const float x = u2f((uint32_t)i);
            const float ref = x / C;
            const float q = x * R;
            const float r = fmaf(-C, q, x);
            const float q2 = fmaf(r, R, q);
            ++tested;
            const int same = (ref != ref && q2 != q2) || f2u(ref) == f2u(q2);
            if (!same) {
                ++ndiff;
                const float ax = fabsf(x);
                if (ax < dmin) dmin = ax;
                if (ax > dmax) dmax = ax;
                if (x == 0.f || (ax >= glo && ax <= ghi)) ++ndiff_guarded;
            }
        }
        printf("C = %.9g (R = %.9g): tested %llu (stride %llu): %llu differ, |x| in [%.6g, %.6g]; inside the guard (x == 0 or 2^-100 <= |x| <= 2^100): %llu\n",
               (double)C, (double)R, (unsigned long long)tested, (unsigned long long)stride, (unsigned long long)ndiff, (double)dmin, (double)dmax,
               (unsigned long long)ndiff_guarded);
        if (ndiff_guarded) bad = 1;
    }
    return bad;
}
