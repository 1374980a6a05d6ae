/*
 * oracle/logenc.c -- CPU restatement of ImProcFunctions::logEncoding (rtengine/iplogenc.cc:38-247,395-402).
 * TEST INFRASTRUCTURE ONLY.  PARITY: xlogf/xexpf/pow_F (sleef) pinned by tests/golden; the function itself unpinned
 * (iplogenc.cc needs glibmm/lcms2 headers: not buildable here).  Highlight compression > 0 (std::pow per pixel) is restated with
 * this host's powf, like the reference would run it.
 */
#include "oracle.h"
#include "oracle_common.h"
#include <float.h>
#include <stdlib.h>

/* find_gray (iplogenc.cc:38-91): bisection on base^source - 1 - target * base + target */
static float fg_f(float x, float source_gray, float target_gray) { return powf(x, source_gray) - 1 - target_gray * x + target_gray; }
float oracle_logenc_find_gray(float source_gray, float target_gray)
{
    if (source_gray <= 0.f) return 0.f;
    float lo = 1.f;
    while (fg_f(lo, source_gray, target_gray) <= 0.f) lo *= 2.f;
    float hi = lo * 2.f;
    while (fg_f(hi, source_gray, target_gray) >= 0.f) hi *= 2.f;
    if (isinf(hi)) return 0.f;
    for (int iter = 0; iter < 100; ++iter) {
        const float mid = lo + (hi - lo) / 2.f;
        const float v = fg_f(mid, source_gray, target_gray);
        if (fabsf(v) < 1e-4f || (hi - lo) / lo <= 1e-4f) return mid;
        if (v > 0.f) lo = mid; else hi = mid;
    }
    return 0.f;
}

typedef struct {
    float gray, shadows_range, dynamic_range, noise, log2, linbase;
    int hlcompr, satcontrol;
    float hlcompr_factor, compr_p, compr_s;
    const double *ws;
} le_t;

/* power_norm / norm (L95-116) */
static float le_norm(float r, float g, float b, const double *ws)
{
    const float hi = FLT_MAX / 100.f;
    const float lum = (float)(r * ws[3] + g * ws[4] + b * ws[5]);
    r = fabsf(r); g = fabsf(g); b = fabsf(b);
    const float r2 = r * r, g2 = g * g, b2 = b * b;
    const float d = r2 + g2 + b2;
    const float n = r * r2 + g * g2 + b * b2;
    const float pn = n / rt_maxf(d, 1e-12f);
    return rt_minf(hi, pn / 2.f + lum / 2.f);
}
static float le_compr(const le_t *p, float x)
{
    const float compr_t = 0.8f;
    if (x < compr_t) return x;
    const float n = (x - compr_t) / p->compr_s;
    const float d = powf(1.f + powf((x - compr_t) / p->compr_s, p->compr_p), 1.f / p->compr_p);
    float res = compr_t + p->compr_s * n / d;
    if (p->hlcompr_factor < 0.1f) res = intpf(p->hlcompr_factor * 10.f, res, x);
    return res;
}
static float le_apply(const le_t *p, float x)
{
    x = rt_maxf(x, p->noise);
    x = rt_maxf(x / p->gray, p->noise);
    if (p->hlcompr) x = le_compr(p, x);
    x = rt_maxf((oracle_xlogf_s(x) / p->log2 - p->shadows_range) / p->dynamic_range, p->noise);
    if (p->linbase > 0.f) x = oracle_xlog2lin(x, p->linbase);
    return x;
}
static float le_sf(const le_t *p, float s, float c) { return c > p->noise ? 1.f - rt_minf(fabsf(s) / c, 1.f) : 0.f; }
static void le_apply_sat(const le_t *p, float *r, float *g, float *b, float f)
{
    const float ll = (float)(*r * p->ws[3] + *g * p->ws[4] + *b * p->ws[5]);
    const float rl = *r - ll, gl = *g - ll, bl = *b - ll;
    const float m = rt_maxf(rt_maxf(le_sf(p, rl, *r), le_sf(p, gl, *g)), le_sf(p, bl, *b));
    const float s = intpf(m, oracle_pow_F(f, 0.3f) * 0.6f + 0.4f, 1.f);
    *r = ll + s * rl; *g = ll + s * gl; *b = ll + s * bl;
}

/* log_encode (L132-316) on contiguous W x H planes; full_width/full_height: ImProcFunctions::full_width/height */
void oracle_log_encoding(float *const img[3], int W, int H, const double ws[9], double gain, double targetGray, double blackEv, double whiteEv,
                         int regularization, int satcontrol, int highlightCompression, int full_width, int full_height)
{
    le_t p;
    p.ws = ws;
    p.gray = powf(2.f, -(float)gain + log2f(0.18f));                     /* ev2gray (L119-122) */
    p.shadows_range = (float)blackEv;
    { const double dr = whiteEv - blackEv; p.dynamic_range = (float)(dr < 0.5 ? 0.5 : dr); }
    p.noise = oracle_pow_F(2.f, -16.f);
    p.log2 = oracle_xlogf_s(2.f);
    const float b = (targetGray > 1 && targetGray < 100 && p.dynamic_range > 0)
        ? oracle_logenc_find_gray((float)(fabs(blackEv) / p.dynamic_range), (float)(targetGray / 100.f)) : 0.f;
    p.linbase = rt_maxf(b, 0.f);
    p.satcontrol = satcontrol;
    p.hlcompr = highlightCompression > 0;
    p.hlcompr_factor = lim01f((float)highlightCompression / 100.f);
    p.compr_p = std_maxf(p.hlcompr_factor, 0.1f);
    p.compr_s = (1.01f - 0.8f) / powf(powf((1.f - 0.8f) / (1.01f - 0.8f), -p.compr_p) - 1.f, 1.f / p.compr_p);
    const size_t n = (size_t)W * H;
    float *R = img[0], *G = img[1], *B = img[2];
    if (regularization == 0) {
#pragma omp parallel for
        for (size_t k = 0; k < n; ++k) {
            float r = R[k], g = G[k], bb = B[k];
            const float m = le_norm(r / 65535.f, g / 65535.f, bb / 65535.f, ws);
            if (m > p.noise) {
                const float mm = le_apply(&p, m);
                const float f = mm / m;
                r *= f; bb *= f; g *= f;
                if (satcontrol && f < 1.f) le_apply_sat(&p, &r, &g, &bb, f);
            }
            R[k] = r; G[k] = g; B[k] = bb;
        }
        return;
    }
    float *Y = (float *)malloc(sizeof(float) * n), *Y2 = (float *)malloc(sizeof(float) * n);
#pragma omp parallel for
    for (size_t k = 0; k < n; ++k) {
        float v = le_norm(R[k], G[k], B[k], ws) / 65535.f;
        v = rt_maxf(1e-5f, rt_minf(v, 128.f));
        Y2[k] = v;
        const float l = oracle_xlogf_s(v);
        const float ll = roundf(l * 20.f) / 20.f;
        Y[k] = oracle_xexpf_s(ll);
    }
    {
        int m1 = full_width > W ? full_width : W, m2 = full_height > H ? full_height : H;
        const float radius = (m1 > m2 ? m1 : m2) / 30.f;
        oracle_guided_filter(Y2, Y, Y, W, H, (int)radius, 0.005f, 0);
    }
    free(Y2);
    const float blend = lim01f((float)regularization / 100.f);
#pragma omp parallel for
    for (size_t k = 0; k < n; ++k) {
        float r = R[k], g = G[k], bb = B[k];
        const float t = Y[k];
        float t2;
        if (t > p.noise && (t2 = le_norm(r / 65535.f, g / 65535.f, bb / 65535.f, ws)) > p.noise) {
            const float c = le_apply(&p, t);
            float f = c / t;
            const float f2 = le_apply(&p, t2) / t2;
            f = intpf(blend, f, f2);
            r *= f; g *= f; bb *= f;
            if (satcontrol && f < 1.f) le_apply_sat(&p, &r, &g, &bb, f);
            R[k] = r; G[k] = g; B[k] = bb;
        }
    }
    free(Y);
}
