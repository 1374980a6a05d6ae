/*
 * oracle/sleef.c -- restatement of the sleef-derived fp32 math the hot path uses
 * (reference: rtengine/sleef.h and rtengine/sleefsseavx.h).
 *
 * TEST INFRASTRUCTURE ONLY.  PARITY PINNED: every function here is checked bit-for-bit against
 * the reference's own headers compiled in place (oracle/_ref, tests/golden/sleef.npz).
 *
 * The scalar and the 4-lane SSE forms are DIFFERENT functions in the last bits and both occur
 * on the path (bulk lanes vs loop tails, e.g. FTblockDN.cc:673-683):
 *   xexpf  scalar  sleef.h:1247-1265      u = s*(s*u+1)+1 ; ldexpkf: x*(u*u)*(u*u)*2^q'
 *   xexpf  vector  sleefsseavx.h:1326-1345 u = 1+((s*s)*u+s) ; vldexpf: (((x*u)*u)*u)*u*2^q'
 *   xlogf  scalar  sleef.h:1198-1221 ; vector sleefsseavx.h:1232-1255 (differ only in ldexp)
 * mlaf/vmlaf are unfused x*y+z (sleef.h:938, helpersse2.h:157-159): build with -ffp-contract=off.
 */
#include "oracle.h"
#include "oracle_common.h"
#include <xmmintrin.h>

#define R_LN2f 1.442695040888963407359924681001892137426645954152985934135449406931f
#define L2Uf 0.693145751953125f
#define L2Lf 1.428606765330187045e-06f

static inline int32_t f2i(float f) { union { float f; int32_t i; } u; u.f = f; return u.i; }
static inline float i2f(int32_t i) { union { float f; int32_t i; } u; u.i = i; return u.f; }
static inline float mla(float x, float y, float z) { return x * y + z; }
/* _mm_cvt_ss2si / _mm_cvtps_epi32: round to nearest even (sleef.h:903-905) */
static inline int rint_i(float x) { return _mm_cvt_ss2si(_mm_set_ss(x)); }

static inline int ilogbp1f(float d)
{
    int m = d < 5.421010862427522E-20f;
    d = m ? 1.8446744073709552E19f * d : d;
    int q = (f2i(d) >> 23) & 0xff;
    return m ? q - (64 + 0x7e) : q - 0x7e;
}
static inline float ldexpk_scalar(float x, int q)
{
    int m = q >> 31;
    m = (((m + q) >> 6) - m) << 4;
    q = q - (m << 2);
    float u = i2f((int32_t)(m + 0x7f) << 23);
    u = u * u;
    x = x * u * u;
    u = i2f((int32_t)(q + 0x7f) << 23);
    return x * u;
}
static inline float ldexpk_vector(float x, int q)
{
    int m = q >> 31;
    m = (((m + q) >> 6) - m) << 4;
    q = q - (m << 2);
    float u = i2f((int32_t)(m + 0x7f) << 23);
    x = (((x * u) * u) * u) * u;
    u = i2f((int32_t)(q + 0x7f) << 23);
    return x * u;
}

float oracle_xexpf_s(float d)
{
    if (d <= -104.0f) return 0.0f;
    int q = rint_i(d * R_LN2f);
    float s = mla((float)q, -L2Uf, d);
    s = mla((float)q, -L2Lf, s);
    float u = 0.00136324646882712841033936f;
    u = mla(u, s, 0.00836596917361021041870117f);
    u = mla(u, s, 0.0416710823774337768554688f);
    u = mla(u, s, 0.166665524244308471679688f);
    u = mla(u, s, 0.499999850988388061523438f);
    u = mla(s, mla(s, u, 1.f), 1.f);
    return ldexpk_scalar(u, q);
}

static inline float xexpf_v_core(float d)
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
    return ldexpk_vector(u, q);
}
float oracle_xexpf_v(float d)
{
    float u = xexpf_v_core(d);
    return (-104.f > d) ? 0.f : u; /* vselfnotzero(vmaskf_gt(-104, d), u) */
}
float oracle_xexpf_v_nocheck(float d) { return xexpf_v_core(d); }

static inline float xlogf_core(float d, int vector)
{
    int e = ilogbp1f(d * 0.7071f);
    float m = vector ? ldexpk_vector(d, -e) : ldexpk_scalar(d, -e);
    float x = vector ? ((-1.0f + m) / (1.0f + m)) : ((m - 1.0f) / (m + 1.0f));
    float x2 = x * x;
    float t = 0.2371599674224853515625f;
    t = mla(t, x2, 0.285279005765914916992188f);
    t = mla(t, x2, 0.400005519390106201171875f);
    t = mla(t, x2, 0.666666567325592041015625f);
    t = mla(t, x2, 2.0f);
    return x * t + 0.693147180559945286226764f * (float)e;
}
float oracle_xlogf_s(float d)
{
    float x = xlogf_core(d, 0);
    if (d == INFINITY) x = INFINITY;
    if (d < 0) x = NAN;
    if (d == 0) x = -INFINITY;
    return x;
}
float oracle_xlogf_v(float d)
{
    float x = xlogf_core(d, 1);
    if (d == INFINITY) x = INFINITY;
    if (0.f > d) x = NAN;
    if (d == 0) x = -INFINITY;
    return x;
}
float oracle_xlogf_v_nocheck(float d) { return xlogf_core(d, 1); }

/* opthelper.h:24  pow_F(a,b) = xexpf(b*xlogf(a)) ; sleef.h:1303-1313 */
/* xcbrtf, sleef.h:966-991 (scalar only on the path) */
float oracle_xcbrtf(float d)
{
    float x, y, q = 1.0f;
    int e, r;
    e = ilogbp1f(d);
    d = ldexpk_scalar(d, -e);
    r = (e + 6144) % 3;
    q = (r == 1) ? 1.2599210498948731647672106f : q;
    q = (r == 2) ? 1.5874010519681994747517056f : q;
    q = ldexpk_scalar(q, (e + 6144) / 3 - 2048);
    q = i2f(f2i(q) ^ (f2i(d) & (int32_t)0x80000000));
    d = i2f(f2i(d) & 0x7fffffff);
    x = -0.601564466953277587890625f;
    x = mla(x, d, 2.8208892345428466796875f);
    x = mla(x, d, -5.532182216644287109375f);
    x = mla(x, d, 5.898262500762939453125f);
    x = mla(x, d, -3.8095417022705078125f);
    x = mla(x, d, 2.2241256237030029296875f);
    y = d * x * x;
    y = (y - (2.0f / 3.0f) * y * (y * x - 1.0f)) * q;
    return y;
}
void oracle_t_xcbrtf(const float *x, float *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = oracle_xcbrtf(x[i]); }

/* atan2kf / xatan2f, sleef.h:1155-1188 (scalar) */
static float atan2kf_(float y, float x)
{
    float s, t, u, q = 0.f;
    if (x < 0) { x = -x; q = -2.f; }
    if (y > x) { t = x; x = y; y = -t; q += 1.f; }
    s = y / x;
    t = s * s;
    u = 0.00282363896258175373077393f;
    u = mla(u, t, -0.0159569028764963150024414f);
    u = mla(u, t, 0.0425049886107444763183594f);
    u = mla(u, t, -0.0748900920152664184570312f);
    u = mla(u, t, 0.106347933411598205566406f);
    u = mla(u, t, -0.142027363181114196777344f);
    u = mla(u, t, 0.199926957488059997558594f);
    u = mla(u, t, -0.333331018686294555664062f);
    t = u * t;
    t = mla(t, s, s);
    return mla(q, (float)1.57079632679489661923, t);
}
static inline float mulsign_(float x, float y) { return i2f(f2i(x) ^ (f2i(y) & (int32_t)0x80000000)); }
static inline int isinf_(float x) { return x == INFINITY || x == -INFINITY; }
float oracle_xatan2f(float y, float x)
{
    const float PI_F = (float)3.14159265358979323846;
    float r = atan2kf_(i2f(f2i(y) & 0x7fffffff), x);
    r = mulsign_(r, x);
    if (isinf_(x) || x == 0) r = PI_F / 2 - (isinf_(x) ? (copysignf(1.f, x) * (float)(PI_F * .5f)) : 0);
    if (isinf_(y)) r = PI_F / 2 - (isinf_(x) ? (copysignf(1.f, x) * (float)(PI_F * .25f)) : 0);
    if (y == 0) r = (copysignf(1.f, x) == -1 ? PI_F : 0);
    return (x != x) || (y != y) ? NAN : mulsign_(r, y);
}
/* xsincosf(float) on SSE2 = lane 0 of the vector form, sleef.h:1048-1052 -> sleefsseavx.h:1051-1100 */
void oracle_xsincosf(float d, float *sn, float *cs)
{
    const int q = rint_i(d * (float)0.63661977236758134308);
    float u = (float)q, s = d, t, rx, ry;
    s = mla(u, -0.78515625f * 2, s);
    s = mla(u, -0.00024127960205078125f * 2, s);
    s = mla(u, -6.3329935073852539062e-07f * 2, s);
    s = mla(u, -4.9604681473525147339e-10f * 2, s);
    t = s;
    s = s * s;
    u = -0.000195169282960705459117889f;
    u = mla(u, s, 0.00833215750753879547119141f);
    u = mla(u, s, -0.166666537523269653320312f);
    u = (u * s) * t;
    rx = t + u;
    u = -2.71811842367242206819355e-07f;
    u = mla(u, s, 2.47990446951007470488548e-05f);
    u = mla(u, s, -0.00138888787478208541870117f);
    u = mla(u, s, 0.0416666641831398010253906f);
    u = mla(u, s, -0.5f);
    ry = 1.f + s * u;
    float x = (q & 1) == 0 ? rx : ry, y = (q & 1) == 0 ? ry : rx;
    if ((q & 2) == 2) x = i2f(f2i(x) ^ (int32_t)0x80000000);
    if (((q + 1) & 2) == 2) y = i2f(f2i(y) ^ (int32_t)0x80000000);
    if (isinf_(d)) x = y = NAN;
    *sn = x; *cs = y;
}
void oracle_t_xatan2f(const float *y, const float *x, float *r, size_t n) { for (size_t i = 0; i < n; ++i) r[i] = oracle_xatan2f(y[i], x[i]); }
void oracle_t_xsincosf(const float *d, float *sn, float *cs, size_t n) { for (size_t i = 0; i < n; ++i) oracle_xsincosf(d[i], sn + i, cs + i); }

float oracle_pow_F(float a, float b) { return oracle_xexpf_s(b * oracle_xlogf_s(a)); }
float oracle_xlin2log(float x, float base) { return oracle_xlogf_s(x * (base - 1.f) + 1.f) / oracle_xlogf_s(base); }
float oracle_xlog2lin(float x, float base) { return (oracle_pow_F(base, x) - 1.f) / (base - 1.f); }

#define MAP1(name, fn) void name(const float *x, float *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = fn(x[i]); }
MAP1(oracle_t_xexpf_s, oracle_xexpf_s)
MAP1(oracle_t_xexpf_v, oracle_xexpf_v)
MAP1(oracle_t_xexpf_vn, oracle_xexpf_v_nocheck)
MAP1(oracle_t_xlogf_s, oracle_xlogf_s)
MAP1(oracle_t_xlogf_v, oracle_xlogf_v)
MAP1(oracle_t_xlogf_vn, oracle_xlogf_v_nocheck)
void oracle_t_pow_F(const float *a, const float *b, float *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = oracle_pow_F(a[i], b[i]); }
void oracle_t_xlin2log(const float *x, float base, float *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = oracle_xlin2log(x[i], base); }
void oracle_t_xlog2lin(const float *x, float base, float *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = oracle_xlog2lin(x[i], base); }


/* ---- double-precision xlog / xexp (rtengine/sleef.h:26-28,58-92,519-571), used by the elementary curves of curves.h:92-156 ---- */
static int64_t d_bits(double d) { int64_t i; memcpy(&i, &d, 8); return i; }
static double d_from_bits(int64_t i) { double d; memcpy(&d, &i, 8); return d; }
static double d_mla(double x, double y, double z) { return x * y + z; }
static double d_ldexpk(double x, int q)
{
    int m = q >> 31;
    m = (((m + q) >> 9) - m) << 7;
    q = q - (m << 2);
    double u = d_from_bits(((int64_t)(m + 0x3ff)) << 52);
    double u2 = u * u;
    u2 = u2 * u2;
    x = x * u2;
    u = d_from_bits(((int64_t)(q + 0x3ff)) << 52);
    return x * u;
}
static int d_ilogbp1(double d)
{
    int m = d < 4.9090934652977266E-91;
    d = m ? 2.037035976334486E90 * d : d;
    int q = (int)((d_bits(d) >> 52) & 0x7ff);
    q = m ? q - (300 + 0x03fe) : q - 0x03fe;
    return q;
}
double oracle_xlog(double d)
{
    int e = d_ilogbp1(d * 0.7071);
    double m = d_ldexpk(d, -e);
    double x = (m - 1) / (m + 1);
    double x2 = x * x;
    double t = 0.148197055177935105296783;
    t = d_mla(t, x2, 0.153108178020442575739679);
    t = d_mla(t, x2, 0.181837339521549679055568);
    t = d_mla(t, x2, 0.22222194152736701733275);
    t = d_mla(t, x2, 0.285714288030134544449368);
    t = d_mla(t, x2, 0.399999999989941956712869);
    t = d_mla(t, x2, 0.666666666666685503450651);
    t = d_mla(t, x2, 2);
    x = x * t + 0.693147180559945286226764 * e;
    if (d == INFINITY) x = INFINITY;
    if (d < 0) x = NAN;
    if (d == 0) x = -INFINITY;
    return x;
}
double oracle_xexp(double d)
{
    double r = d * 1.442695040888963407359924681001892137426645954152985934135449406931;
    int q = (int)(r < 0 ? (int)(r - 0.5) : (int)(r + 0.5));
    double s = d_mla(q, -.69314718055966295651160180568695068359375, d);
    s = d_mla(q, -.28235290563031577122588448175013436025525412068e-12, s);
    double u = 2.08860621107283687536341e-09;
    u = d_mla(u, s, 2.51112930892876518610661e-08);
    u = d_mla(u, s, 2.75573911234900471893338e-07);
    u = d_mla(u, s, 2.75572362911928827629423e-06);
    u = d_mla(u, s, 2.4801587159235472998791e-05);
    u = d_mla(u, s, 0.000198412698960509205564975);
    u = d_mla(u, s, 0.00138888888889774492207962);
    u = d_mla(u, s, 0.00833333333331652721664984);
    u = d_mla(u, s, 0.0416666666666665047591422);
    u = d_mla(u, s, 0.166666666666666851703837);
    u = d_mla(u, s, 0.5);
    u = s * s * u + s + 1;
    u = d_ldexpk(u, q);
    if (d == -INFINITY) u = 0;
    return u;
}
void oracle_t_xlog(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = oracle_xlog(x[i]); }
void oracle_t_xexp(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = oracle_xexp(x[i]); }
