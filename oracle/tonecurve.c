/*
 * oracle/tonecurve.c -- CPU oracle for the NEUTRAL tone-curve mode (ART's default, procparams.cc:1585).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * Restates
 *   NeutralToneCurve::ApplyState ctor / BatchApply    rtengine/curves.cc:854-1038   (basecurve == nullptr: BcMode::LINEAR)
 *   Color::rgb2jzczhz / jzczhz2rgb and friends         rtengine/color.h:1761-1805, rtengine/color.cc:6690-6742
 *   XYZ_D50_to_D65 / XYZ_D65_to_D50 / PQ / PQ_inv      rtengine/color.cc:37-86, PQ LUTs color.cc:323-326 (flags 0)
 *   Color::filmlike_clip                                rtengine/color.cc:6650-6688
 *   curves::setLutVal                                   rtengine/curves.h:224-231
 *   dot_product(Mat33, Vec3)                            rtengine/linalgebra.h:226-239 (accumulates from 0)
 *
 * Third-party arithmetic: PQ()/PQ_inv() call std::pow(float, float) = the host libm's powf (glibc 2.35 here).  The
 * 65536-entry PQ LUTs are built with it on the host, exactly as the reference does at start-up, and serve every pixel
 * whose LMS response lies in [0, 1]; such pixels are bit-exact.  LMS > 1 (values above the PQ LUT, i.e. super-white
 * input) calls powf per pixel: those pixels are flagged in `out_of_lut_range` and compared with a tolerance.
 * Parity: xatan2f / xsincosf are pinned against oracle/_ref (tests/golden/sleef2.npz); the BatchApply body itself is
 * PARITY UNPINNED (curves.cc needs glibmm/lcms2 to compile).
 */
#include "oracle.h"
#include "oracle_common.h"
#include <stdlib.h>

static float PQ(float X)
{
    X = std_maxf(X, 1e-10f);
    const float XX = powf(X * 1e-4f, 0.1593017578125f);
    return powf((0.8359375f + 18.8515625f * XX) / (1 + 18.6875f * XX), 134.034375f);
}
static float PQ_inv(float X)
{
    X = std_maxf(X, 1e-10f);
    const float XX = powf(X, 7.460772656268214e-03f);
    return 1e4f * powf((0.8359375f - XX) / (18.6875f * XX - 18.8515625f), 6.277394636015326f);
}
void oracle_pq_luts(float *pq, float *pq_inv)
{
    for (int i = 0; i < 65536; ++i) {
        pq[i] = PQ((float)i / 65535.f);
        pq_inv[i] = PQ_inv((float)i / 65535.f);
    }
}
/* LUTf::operator[](float) with flags 0, for index >= 0 */
static inline float lut_noclip(const float *data, float index)
{
    const int maxs = 65534;
    int idx = (int)index;
    if (index < 0.f || !(index == index)) idx = 0;
    else if (index > (float)maxs) idx = maxs;
    const float diff = index - (float)idx;
    const float p1 = data[idx], p2 = data[idx + 1] - p1;
    return p1 + p2 * diff;
}
static inline void mat_vec(const float m[9], const float v[3], float r[3])
{
    for (int i = 0; i < 3; ++i) {
        float acc = 0;
        for (int k = 0; k < 3; ++k) acc += m[3 * i + k] * v[k];
        r[i] = acc;
    }
}
/* test export: n products m[k] (9 floats) x v[k] (3 floats), pinned against linalgebra.h's dot_product (tests/test_golden_pins.py) */
void oracle_t_mat_vec(const float *m, const float *v, float *r, size_t n)
{
    for (size_t k = 0; k < n; ++k) mat_vec(m + 9 * k, v + 3 * k, r + 3 * k);
}
static const float D50_D65[9] = {0.9555766f, -0.0230393f, 0.0631636f, -0.0282895f, 1.0099416f, 0.0210077f, 0.0122982f, -0.0204830f, 1.3299098f};
static const float D65_D50[9] = {1.0478112f, 0.0228866f, -0.0501270f, 0.0295424f, 0.9904844f, -0.0170491f, -0.0092345f, 0.0150436f, 0.7521316f};

typedef struct { const float *pq, *pq_inv; int oor; } pqctx;

static float get_pq(pqctx *c, float x)
{
    if (x >= 0.f && x <= 1.f) return lut_noclip(c->pq, x * 65535.f);
    if (x > 1.f) c->oor = 1;        /* x < 0 (or NaN) evaluates the constant PQ(1e-10f) */
    return PQ(x);
}
static float get_pq_inv(pqctx *c, float x)
{
    if (x >= 0.f && x <= 1.f) return lut_noclip(c->pq_inv, x * 65535.f);
    if (x > 1.f) c->oor = 1;
    return PQ_inv(x);
}
static void rgb2jzczhz(pqctx *c, float R, float G, float B, float *Jz, float *cz, float *hz, const float ws[9])
{
    float v[3] = {ws[0] * R + ws[1] * G + ws[2] * B, ws[3] * R + ws[4] * G + ws[5] * B, ws[6] * R + ws[7] * G + ws[8] * B}, d[3];
    mat_vec(D50_D65, v, d);
    const float X = d[0], Y = d[1], Z = d[2];
    const float Lp = get_pq(c, 0.674207838f * X + 0.382799340f * Y - 0.047570458f * Z);
    const float Mp = get_pq(c, 0.149284160f * X + 0.739628340f * Y + 0.083327300f * Z);
    const float Sp = get_pq(c, 0.070941080f * X + 0.174768000f * Y + 0.670970020f * Z);
    const float Iz = 0.5f * (Lp + Mp);
    const float az = 3.524000f * Lp - 4.066708f * Mp + 0.542708f * Sp;
    const float bz = 0.199076f * Lp + 1.096799f * Mp - 1.295875f * Sp;
    *Jz = (0.44f * Iz) / (1.f - 0.56f * Iz) - 1.6295499532821566e-11f;
    *cz = sqrtf(sqrf(bz) + sqrf(az));       /* yuv2hsl(u = bz, v = az) */
    *hz = oracle_xatan2f(bz, az);
}
static void jzczhz2rgb(pqctx *c, float Jz, float cz, float hz, float *R, float *G, float *B, const float iws[9])
{
    float sn, cs;
    oracle_xsincosf(hz, &sn, &cs);
    const float bz = cz * sn, az = cz * cs; /* hsl2yuv(h, s, u = bz, v = az) */
    Jz = Jz + 1.6295499532821566e-11f;
    const float Iz = Jz / (0.44f + 0.56f * Jz);
    const float L = get_pq_inv(c, Iz + 1.386050432715393e-1f * az + 5.804731615611869e-2f * bz);
    const float M = get_pq_inv(c, Iz - 1.386050432715393e-1f * az - 5.804731615611891e-2f * bz);
    const float S = get_pq_inv(c, Iz - 9.601924202631895e-2f * az - 8.118918960560390e-1f * bz);
    float v[3], d[3];
    v[0] = +1.661373055774069e+00f * L - 9.145230923250668e-01f * M + 2.313620767186147e-01f * S;
    v[1] = -3.250758740427037e-01f * L + 1.571847038366936e+00f * M - 2.182538318672940e-01f * S;
    v[2] = -9.098281098284756e-02f * L - 3.127282905230740e-01f * M + 1.522766561305260e+00f * S;
    mat_vec(D65_D50, v, d);
    *R = iws[0] * d[0] + iws[1] * d[1] + iws[2] * d[2];
    *G = iws[3] * d[0] + iws[4] * d[1] + iws[5] * d[2];
    *B = iws[6] * d[0] + iws[7] * d[1] + iws[8] * d[2];
}

static void clip_tone(float *r, float *g, float *b, const float L)
{
    const float r_ = *r > L ? L : *r;
    const float b_ = *b > L ? L : *b;
    const float g_ = b_ + ((r_ - b_) * (*g - *b) / (*r - *b));
    *r = r_; *g = g_; *b = b_;
}
static void filmlike_clip(float *r, float *g, float *b, float L)
{
    if (*r >= *g) {
        if (*g > *b) clip_tone(r, g, b, L);
        else if (*b > *r) clip_tone(b, r, g, L);
        else if (*b > *g) clip_tone(r, b, g, L);
        else { *r = *r > L ? L : *r; *g = *g > L ? L : *g; *b = *g; }
    } else {
        if (*r >= *b) clip_tone(g, r, b, L);
        else if (*b > *g) clip_tone(b, g, r, L);
        else clip_tone(g, b, r, L);
    }
}

static float *g_pq, *g_pq_inv;
static void ensure_luts(void)
{
    if (!g_pq) {
        float *a = (float *)malloc(sizeof(float) * 65536), *b = (float *)malloc(sizeof(float) * 65536);
        oracle_pq_luts(a, b);
        g_pq_inv = b; g_pq = a;
    }
}

void oracle_neutral_state_init(oracle_neutral_state *st, const double ws[9], const double iws[9], const float *to_out, const float *to_work)
{
    static const float hws[9] = {0.6734241f, 0.1656411f, 0.1251286f, 0.2790177f, 0.6753402f, 0.0456377f, -0.0019300f, 0.0299784f, 0.7973330f};
    static const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    ensure_luts();
    pqctx c = {g_pq, g_pq_inv, 0};
    for (int k = 0; k < 9; ++k) {
        st->ws[k] = (float)ws[k]; st->iws[k] = (float)iws[k];
        st->to_out[k] = to_out ? to_out[k] : ident[k]; st->to_work[k] = to_work ? to_work[k] : ident[k];
    }
    float j, ch, ohue;
    rgb2jzczhz(&c, 1, 0, 0, &j, &ch, &st->rhue, hws);
    rgb2jzczhz(&c, 0, 0, 1, &j, &ch, &st->bhue, hws);
    rgb2jzczhz(&c, 1, 1, 0, &j, &ch, &st->yhue, hws);
    rgb2jzczhz(&c, 1, 0.5f, 0, &j, &ch, &ohue, hws);
    st->yrange = fabsf(ohue - st->yhue) * 0.8f;
    st->rrange = fabsf(ohue - st->rhue);
    st->brange = st->rrange;
}

static inline float gauss(float x, float b, float c) { return oracle_xexpf_s(-sqrf(x - b) / (2 * sqrf(c))); }
static inline float lim_f(float a, float lo, float hi) { return rt_maxf(lo, rt_minf(a, hi)); }

/* whitecoeff = ToneCurve::whitecoeff (whitept in BatchApply = 65535*whitecoeff, Lmax = that whitept: curves.cc:896,
 * 221-225).  out_of_lut_range (w*h bytes, may be NULL) marks pixels that needed powf per pixel. */
void oracle_tone_curve_neutral(float *const img[3], size_t s, int w, int h, const float *lut, float whitecoeff,
                               const oracle_neutral_state *st, unsigned char *oor)
{
    ensure_luts();
    const float whitept = 65535.f * whitecoeff;
    const float Lmax = whitept;
    static const float dl[3] = {1.1f, 1.2f, 1.5f}, th[3] = {0.85f, 0.75f, 0.95f};
    float sc[3];
    for (int i = 0; i < 3; ++i) sc[i] = (1.f - th[i]) / sqrtf(dl[i] - 1.f);
    const float PI_180 = (float)(3.14159265358979323846 / 180.0);
#pragma omp parallel for
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t o = (size_t)y * s + x;
            pqctx c = {g_pq, g_pq_inv, 0};
            float rgb[3], jch[3], t[3];
            rgb[0] = std_maxf(img[0][o] / 65535.f, 0.f);
            rgb[1] = std_maxf(img[1][o] / 65535.f, 0.f);
            rgb[2] = std_maxf(img[2][o] / 65535.f, 0.f);
            rgb2jzczhz(&c, rgb[0], rgb[1], rgb[2], &jch[0], &jch[1], &jch[2], st->ws);
            const float ilum = jch[0];
            float hue = jch[2];
            const float iY = (rgb[0] + rgb[1] + rgb[2]) / 3.f;
            mat_vec(st->to_out, rgb, t); rgb[0] = t[0]; rgb[1] = t[1]; rgb[2] = t[2];
            const float ac = rt_maxf(rt_maxf(rgb[0], rgb[1]), rgb[2]);     /* max(a,b,c), rt_math.h:78-88 */
            float d[3] = {0.f, 0.f, 0.f};
            const float aac = fabsf(ac);
            if (ac != 0.f) {
                d[0] = (ac - rgb[0]) / aac;
                d[1] = (ac - rgb[1]) / aac;
                d[2] = (ac - rgb[2]) / aac;
            }
            float cd[3];
            for (int i = 0; i < 3; ++i)
                cd[i] = d[i] < th[i] ? d[i] : sc[i] * sqrtf(d[i] - th[i] + sqrf(sc[i]) / 4.0f) - sc[i] * sqrtf(sqrf(sc[i]) / 4.0f) + th[i];
            rgb[0] = ac - cd[0] * aac;
            rgb[1] = ac - cd[1] * aac;
            rgb[2] = ac - cd[2] * aac;
            mat_vec(st->to_work, rgb, t); rgb[0] = t[0]; rgb[1] = t[1]; rgb[2] = t[2];
            const float oY = (rgb[0] + rgb[1] + rgb[2]) / 3.f;
            if (oY > 0.f) {
                const float f = iY / oY;
                rgb[0] *= f; rgb[1] *= f; rgb[2] *= f;
                filmlike_clip(&rgb[0], &rgb[1], &rgb[2], Lmax);
            }
            for (int j = 0; j < 3; ++j) {
                float nt = rgb[j] * 65535.f;
                nt = oracle_set_lut_val(lut, nt);      /* curves::setLutVal (curves.h:224-231) */
                rgb[j] = nt / 65535.f;
            }
            rgb2jzczhz(&c, rgb[0], rgb[1], rgb[2], &jch[0], &jch[1], &jch[2], st->ws);
            float hue_shift = 15.f * PI_180 * gauss(hue, st->rhue, st->rrange);
            hue_shift += -5.f * PI_180 * gauss(hue, st->bhue, st->brange);
            hue_shift *= lim01f((rgb[0] + rgb[1] + rgb[2]) / (3.f * whitecoeff));
            hue += hue_shift;
            float sat = jch[1];
            {
                const float olum = jch[0];
                float ccf = ilum > 1e-5f ? (1.f - (lim01f((olum / ilum) - 1.f) * 0.2f)) : 1.f;
                ccf = lim01f(ccf + 0.5f * gauss(hue, st->yhue, st->yrange));
                sat *= ccf;
            }
            jzczhz2rgb(&c, jch[0], sat, hue, &rgb[0], &rgb[1], &rgb[2], st->iws);
            img[0][o] = lim_f(rgb[0] * 65535.f, 0.f, whitept);
            img[1][o] = lim_f(rgb[1] * 65535.f, 0.f, whitept);
            img[2][o] = lim_f(rgb[2] * 65535.f, 0.f, whitept);
            if (oor) oor[(size_t)y * w + x] = (unsigned char)c.oor;
        }
}
