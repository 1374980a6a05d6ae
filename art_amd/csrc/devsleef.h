// art_amd/csrc/devsleef.h -- device-side sleef-derived fp32 math, scalar (_s) and per-lane SSE (_v)
// forms (reference: rtengine/sleef.h:938-966,1198-1265 and rtengine/sleefsseavx.h:978-1000,1232-1372).
// Both forms occur on the path (bulk lanes vs loop tails) and differ in the last bits:
//   exp: scalar  u = s*(s*u+1)+1, ldexp x*(u*u)*(u*u)*2^q' ; vector u = 1+((s*s)*u+s), ldexp (((x*u)*u)*u)*u*2^q'
// mlaf/vmlaf are unfused x*y+z: this file must be compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace artgpu {

#define ART_R_LN2f 1.442695040888963407359924681001892137426645954152985934135449406931f
#define ART_L2Uf 0.693145751953125f
#define ART_L2Lf 1.428606765330187045e-06f

__device__ __forceinline__ float sl_mla(float x, float y, float z) { return x * y + z; }

__device__ __forceinline__ int sl_ilogbp1f(float d)
{
    const bool m = d < 5.421010862427522E-20f;
    d = m ? 1.8446744073709552E19f * d : d;
    const int q = (__float_as_int(d) >> 23) & 0xff;
    return m ? q - (64 + 0x7e) : q - 0x7e;
}
template <bool VEC>
__device__ __forceinline__ float sl_ldexpk(float x, int q)
{
    int m = q >> 31;
    m = (((m + q) >> 6) - m) << 4;
    q = q - (m << 2);
    float u = __int_as_float((m + 0x7f) << 23);
    if (VEC) {
        x = (((x * u) * u) * u) * u;
    } else {
        u = u * u;
        x = x * u * u;
    }
    u = __int_as_float((q + 0x7f) << 23);
    return x * u;
}
template <bool VEC>
__device__ __forceinline__ float sl_exp_core(float d)
{
    const int q = __float2int_rn(d * ART_R_LN2f); // cvtss2si / cvtps2dq: round to nearest even
    float s = sl_mla((float)q, -ART_L2Uf, d);
    s = sl_mla((float)q, -ART_L2Lf, s);
    float u = 0.00136324646882712841033936f;
    u = sl_mla(u, s, 0.00836596917361021041870117f);
    u = sl_mla(u, s, 0.0416710823774337768554688f);
    u = sl_mla(u, s, 0.166665524244308471679688f);
    u = sl_mla(u, s, 0.499999850988388061523438f);
    if (VEC) u = 1.0f + sl_mla(s * s, u, s);
    else u = sl_mla(s, sl_mla(s, u, 1.f), 1.f);
    return sl_ldexpk<VEC>(u, q);
}
__device__ __forceinline__ float xexpf_s(float d) { return d <= -104.0f ? 0.0f : sl_exp_core<false>(d); }
__device__ __forceinline__ float xexpf_v(float d) { const float u = sl_exp_core<true>(d); return (-104.f > d) ? 0.f : u; }
__device__ __forceinline__ float xexpf_v_nocheck(float d) { return sl_exp_core<true>(d); }
// xexpf_v with sl_ldexpk's five multiplications by powers of two as ONE v_ldexp_f32.  Multiplying by a power of two is exact until the value
// leaves the normal range, so the two agree wherever exp(d) is a normal number; where it is subnormal (d < -87.3) ldexpk rounds more than
// once and the results can differ in the last bits of a number below 1.2e-38 (scripts/exp_ldexp_check.c walks every float and reports exactly
// that set).  For callers that add the result to something at least 2^24 times larger -- see the call sites.
__device__ __forceinline__ float xexpf_v_ldexp(float d)
{
    const float qf = __builtin_rintf(d * ART_R_LN2f);      // (float)q without the way through the integer: the same number while q fits an int
    const int q = (int)qf;
    float s = sl_mla(qf, -ART_L2Uf, d);
    s = sl_mla(qf, -ART_L2Lf, s);
    float u = 0.00136324646882712841033936f;
    u = sl_mla(u, s, 0.00836596917361021041870117f);
    u = sl_mla(u, s, 0.0416710823774337768554688f);
    u = sl_mla(u, s, 0.166665524244308471679688f);
    u = sl_mla(u, s, 0.499999850988388061523438f);
    u = 1.0f + sl_mla(s * s, u, s);
    u = __builtin_amdgcn_ldexpf(u, q);
    return (-104.f > d) ? 0.f : u;
}

template <bool VEC>
__device__ __forceinline__ float sl_log_core(float d)
{
    const int e = sl_ilogbp1f(d * 0.7071f);
    const float m = sl_ldexpk<VEC>(d, -e);
    const float x = VEC ? ((-1.0f + m) / (1.0f + m)) : ((m - 1.0f) / (m + 1.0f));
    const float x2 = x * x;
    float t = 0.2371599674224853515625f;
    t = sl_mla(t, x2, 0.285279005765914916992188f);
    t = sl_mla(t, x2, 0.400005519390106201171875f);
    t = sl_mla(t, x2, 0.666666567325592041015625f);
    t = sl_mla(t, x2, 2.0f);
    return x * t + 0.693147180559945286226764f * (float)e;
}
template <bool VEC>
__device__ __forceinline__ float sl_logf(float d)
{
    float x = sl_log_core<VEC>(d);
    if (d == INFINITY) x = INFINITY;
    if (d < 0.f) x = NAN;
    if (d == 0.f) x = -INFINITY;
    return x;
}
__device__ __forceinline__ float xlogf_s(float d) { return sl_logf<false>(d); }
__device__ __forceinline__ float xlogf_v(float d) { return sl_logf<true>(d); }
__device__ __forceinline__ float xlogf_v_nocheck(float d) { return sl_log_core<true>(d); }

// xcbrtf (rtengine/sleef.h:966-991), scalar form
__device__ __forceinline__ float xcbrtf_s(float d)
{
    float q = 1.0f;
    const int e = sl_ilogbp1f(d);
    d = sl_ldexpk<false>(d, -e);
    const int r = (e + 6144) % 3;
    q = (r == 1) ? 1.2599210498948731647672106f : q;
    q = (r == 2) ? 1.5874010519681994747517056f : q;
    q = sl_ldexpk<false>(q, (e + 6144) / 3 - 2048);
    q = __int_as_float(__float_as_int(q) ^ (__float_as_int(d) & (int)0x80000000));
    d = __int_as_float(__float_as_int(d) & 0x7fffffff);
    float x = -0.601564466953277587890625f;
    x = sl_mla(x, d, 2.8208892345428466796875f);
    x = sl_mla(x, d, -5.532182216644287109375f);
    x = sl_mla(x, d, 5.898262500762939453125f);
    x = sl_mla(x, d, -3.8095417022705078125f);
    x = sl_mla(x, d, 2.2241256237030029296875f);
    float y = d * x * x;
    y = (y - (2.0f / 3.0f) * y * (y * x - 1.0f)) * q;
    return y;
}

// xatan2f (rtengine/sleef.h:1155-1188), scalar form
__device__ __forceinline__ float sl_atan2kf(float y, float x)
{
    float q = 0.f;
    if (x < 0) { x = -x; q = -2.f; }
    if (y > x) { const float t = x; x = y; y = -t; q += 1.f; }
    const float s = y / x;
    float t = s * s;
    float u = 0.00282363896258175373077393f;
    u = sl_mla(u, t, -0.0159569028764963150024414f);
    u = sl_mla(u, t, 0.0425049886107444763183594f);
    u = sl_mla(u, t, -0.0748900920152664184570312f);
    u = sl_mla(u, t, 0.106347933411598205566406f);
    u = sl_mla(u, t, -0.142027363181114196777344f);
    u = sl_mla(u, t, 0.199926957488059997558594f);
    u = sl_mla(u, t, -0.333331018686294555664062f);
    t = u * t;
    t = sl_mla(t, s, s);
    return sl_mla(q, (float)1.57079632679489661923, t);
}
__device__ __forceinline__ float sl_mulsign(float x, float y) { return __int_as_float(__float_as_int(x) ^ (__float_as_int(y) & (int)0x80000000)); }
__device__ __forceinline__ bool sl_isinf(float x) { return x == __builtin_huge_valf() || x == -__builtin_huge_valf(); }
__device__ __forceinline__ float xatan2f_s(float y, float x)
{
    const float PI_F = (float)3.14159265358979323846;
    float r = sl_atan2kf(__int_as_float(__float_as_int(y) & 0x7fffffff), x);
    r = sl_mulsign(r, x);
    const float sgx = copysignf(1.f, x);
    if (sl_isinf(x) || x == 0) r = PI_F / 2 - (sl_isinf(x) ? (sgx * (float)(PI_F * .5f)) : 0);
    if (sl_isinf(y)) r = PI_F / 2 - (sl_isinf(x) ? (sgx * (float)(PI_F * .25f)) : 0);
    if (y == 0) r = (sgx == -1 ? PI_F : 0);
    return (x != x) || (y != y) ? __builtin_nanf("") : sl_mulsign(r, y);
}
// xsincosf(float) under SSE2 = lane 0 of the vector form (sleef.h:1048-1052 -> sleefsseavx.h:1051-1100)
__device__ __forceinline__ void xsincosf_v(float d, float &sn, float &cs)
{
    const int q = __float2int_rn(d * (float)0.63661977236758134308);
    float u = (float)q, s = d;
    s = sl_mla(u, -0.78515625f * 2, s);
    s = sl_mla(u, -0.00024127960205078125f * 2, s);
    s = sl_mla(u, -6.3329935073852539062e-07f * 2, s);
    s = sl_mla(u, -4.9604681473525147339e-10f * 2, s);
    const float t = s;
    s = s * s;
    u = -0.000195169282960705459117889f;
    u = sl_mla(u, s, 0.00833215750753879547119141f);
    u = sl_mla(u, s, -0.166666537523269653320312f);
    u = (u * s) * t;
    const float rx = t + u;
    u = -2.71811842367242206819355e-07f;
    u = sl_mla(u, s, 2.47990446951007470488548e-05f);
    u = sl_mla(u, s, -0.00138888787478208541870117f);
    u = sl_mla(u, s, 0.0416666641831398010253906f);
    u = sl_mla(u, s, -0.5f);
    const float ry = 1.f + s * u;
    float x = (q & 1) == 0 ? rx : ry, y = (q & 1) == 0 ? ry : rx;
    if ((q & 2) == 2) x = -x;
    if (((q + 1) & 2) == 2) y = -y;
    if (sl_isinf(d)) x = y = __builtin_nanf("");
    sn = x; cs = y;
}

// pow_F, xlin2log, xlog2lin (rtengine/sleef.h:1296-1313): scalar exp / log forms
__device__ __forceinline__ float pow_F(float a, float b) { return xexpf_s(b * xlogf_s(a)); }
__device__ __forceinline__ float xlin2log(float x, float base) { return xlogf_s(x * (base - 1.f) + 1.f) / xlogf_s(base); }
__device__ __forceinline__ float xlog2lin(float x, float base) { return (pow_F(base, x) - 1.f) / (base - 1.f); }

// vclampf (sleefsseavx.h:1396-1399): low for NaN -- vminf / vmaxf are the SSE forms (second operand when unordered)
__device__ __forceinline__ float vclampf(float v, float lo, float hi)
{
    const float m = hi < v ? hi : v;      // vminf(hi, v)
    return m > lo ? m : lo;               // vmaxf(m, lo)
}
// LUTf::operator[](vfloat) (rtengine/LUT.h:349-377), one lane
__device__ __forceinline__ float lutf_vlookup(const float *__restrict__ data, int size, float index)
{
    const int idx = (int)vclampf(index, 0.f, (float)(size - 2));
    const float lower = data[idx], upper = data[idx + 1];
    const float diff = vclampf(index, 0.f, (float)(size - 1)) - (float)idx;
    return diff * upper + (1.f - diff) * lower;
}

// LUTf::operator[](float) (rtengine/LUT.h:436-459): clip_above selects LUT_CLIP_ABOVE behaviour
template <bool CLIP_ABOVE>
__device__ __forceinline__ float lutf_lookup(const float *__restrict__ data, int size, float index)
{
    const int maxs = size - 2;
    if (index < 0.f || !(index == index)) return data[0];
    int idx = (int)index;
    if (index > (float)maxs) {
        if (CLIP_ABOVE) return data[size - 1];
        idx = maxs;
    }
    const float diff = index - (float)idx;
    const float p1 = data[idx];
    const float p2 = data[idx + 1] - p1;
    return p1 + p2 * diff;
}

// The same lookup with entries [0, LUT_LDS_N) of the table resident in LDS (`lds`), the rest read from `data` as before.  A 65536-entry
// table (256 KB) does not fit L1, so each lookup of a streaming kernel is otherwise an L2 line gather; linear scene data sit mostly in
// the lower part of the range.  159 KB of the CU's 160 KB LDS: one workgroup per CU.
constexpr int LUT_LDS_N = 40704;
// (the LDS copy is addressed through an LDS-qualified pointer: with a generic one the compiler merges the two arms below into a select
// of POINTERS and every lookup, LDS or not, becomes a flat_load -- SQ_INSTS_LDS of these kernels was the table fill and nothing else)
typedef __attribute__((address_space(3))) const float lds_cfloat;
template <bool CLIP_ABOVE>
__device__ __forceinline__ float lutf_lookup_lds(const float *lds_generic, const float *__restrict__ data, int size, float index, int lo = 0)
{
    // entries [lo, lo + LUT_LDS_N) of the table are in LDS (lo a multiple of 4)
    lds_cfloat *lds = (lds_cfloat *)lds_generic;
    const int maxs = size - 2;
    if (index < 0.f || !(index == index)) return lo ? data[0] : lds[0];
    int idx = (int)index;
    if (index > (float)maxs) {
        if (CLIP_ABOVE) return data[size - 1];
        idx = maxs;
    }
    const float diff = index - (float)idx;
    float p1, q;
    if ((unsigned)(idx - lo) < (unsigned)(LUT_LDS_N - 1)) { p1 = lds[idx - lo]; q = lds[idx - lo + 1]; }
    else { p1 = data[idx]; q = data[idx + 1]; }
    const float p2 = q - p1;
    return p1 + p2 * diff;
}
__device__ __forceinline__ void lut_lds_fill(float *lds, const float *__restrict__ data, int nthreads, int lo = 0)
{
    const float4 *src = reinterpret_cast<const float4 *>(data + lo);
    float4 *dst = reinterpret_cast<float4 *>(lds);
    for (int i = threadIdx.x; i < LUT_LDS_N / 4; i += nthreads) dst[i] = src[i];
    __syncthreads();
}

} // namespace artgpu
