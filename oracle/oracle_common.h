/*
 * oracle/oracle_common.h -- shared scalar helpers for the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Every helper restates one fp32 primitive of the reference's x86-64 (SSE2) code
 * path, so that a plain-C loop reproduces the reference's bit patterns when it is
 * compiled with -ffp-contract=off (the reference's own flag, CMakeLists.txt:35-38):
 *
 *   sse_minf/sse_maxf : _mm_min_ps/_mm_max_ps operand-order semantics
 *                       (rtengine/helpersse2.h:168-179) -- "returns y if x is NaN"
 *   intpf             : a*b + (1-a)*c, unfused (rtengine/rt_math.h:109-118,
 *                       rtengine/sleefsseavx.h:1435-1442)
 *   median3f          : max(min(a,b), min(c, max(a,b))) (rtengine/median.h:52-64)
 *   xdiv2f/xmul2f/xdivf: exponent-field arithmetic (rtengine/sleef.h:1267-1301)
 *   fc()              : RawImage::FC (rtengine/rawimage.h:186-189)
 */
#ifndef ART_ORACLE_COMMON_H
#define ART_ORACLE_COMMON_H

#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <math.h>

#ifdef __cplusplus
extern "C" {
#endif

static inline unsigned fc(unsigned filters, unsigned row, unsigned col)
{
    return (filters >> (((((row) << 1) & 14) + ((col) & 1)) << 1)) & 3;
}

static inline float sse_minf(float x, float y) { return x < y ? x : y; }
static inline float sse_maxf(float x, float y) { return x > y ? x : y; }
/* rt_math.h:46-76  min(a,b) = b<a ? b : a ; max(a,b) = a<b ? b : a */
static inline float rt_minf(float a, float b) { return b < a ? b : a; }
static inline float rt_maxf(float a, float b) { return a < b ? b : a; }
static inline float sqrf(float x) { return x * x; }
static inline double sqr_d(double x) { return x * x; }
static inline float intpf(float a, float b, float c) { return a * b + (1.f - a) * c; }
static inline float median3_sse(float a, float b, float c)
{
    return sse_maxf(sse_minf(a, b), sse_minf(c, sse_maxf(a, b)));
}
/* std::max(std::min(a,b), std::min(c, std::max(a,b))) with std:: semantics */
static inline float std_minf(float a, float b) { return b < a ? b : a; }
static inline float std_maxf(float a, float b) { return a < b ? b : a; }
static inline float median3_std(float a, float b, float c)
{
    return std_maxf(std_minf(a, b), std_minf(c, std_maxf(a, b)));
}
static inline float lim01f(float a) { return rt_maxf(0.f, rt_minf(a, 1.f)); }

static inline float xmul2f(float d)
{
    union { float f; int32_t i; } u; u.f = d;
    if (u.i & 0x7FFFFFFF) u.i += 1 << 23;
    return u.f;
}
static inline float xdiv2f(float d)
{
    union { float f; int32_t i; } u; u.f = d;
    if (u.i & 0x7FFFFFFF) u.i -= 1 << 23;
    return u.f;
}
static inline float xdivf(float d, int n)
{
    union { float f; int32_t i; } u; u.f = d;
    if (u.i & 0x7FFFFFFF) u.i -= n << 23;
    return u.f;
}

#ifdef __cplusplus
}
#endif
#endif
