/* oracle/helpers_export.c -- array wrappers around the inline helpers of oracle_common.h so that
 * tests can pin them against the reference's own headers (oracle/_ref, tests/golden/helpers.npz).
 * TEST INFRASTRUCTURE ONLY. */
#include "oracle.h"
#include "oracle_common.h"

void oracle_t_minmax(const float *a, const float *b, float *mn, float *mx, size_t n)
{
    for (size_t i = 0; i < n; ++i) { mn[i] = sse_minf(a[i], b[i]); mx[i] = sse_maxf(a[i], b[i]); }
}
void oracle_t_median3(const float *a, const float *b, const float *c, float *y, size_t n)
{
    for (size_t i = 0; i < n; ++i) y[i] = median3_sse(a[i], b[i], c[i]);
}
void oracle_t_intp(const float *a, const float *b, const float *c, float *y, size_t n)
{
    for (size_t i = 0; i < n; ++i) y[i] = intpf(a[i], b[i], c[i]);
}
void oracle_t_xdiv2f(const float *x, float *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = xdiv2f(x[i]); }
void oracle_t_xdivf2(const float *x, float *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = xdivf(x[i], 2); }
