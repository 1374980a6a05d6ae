/*
 * oracle/pixelops.c -- CPU restatement of the per-pixel stages around the demosaic:
 *   oracle_get_image           RawImageSource::getImage, skip=1/tran=0/no highlight recovery
 *                              (reference: rtengine/rawimagesource.cc:781-1104, loop L940-1025)
 *   oracle_convert_color_space colorSpaceConversion_ matrix branch (rawimagesource.cc:3184-3213)
 *   oracle_exposure            ImProcFunctions::expcomp (rtengine/ipexposure.cc:28-72)
 *   oracle_filmlike_clip       filmlike_clip -> Color::filmlike_clip (iptonecurve.cc:214-231,
 *                              color.cc:6648-6690)
 *   oracle_tone_curve_std      StandardToneCurve::Apply via curves::setLutVal and the scalar
 *                              LUTf::operator[](float) (curves.h:224-231,360-368; LUT.h:436-459)
 *
 * TEST INFRASTRUCTURE ONLY.  PARITY: the LUTf lookup is pinned against the reference's own
 * LUT.h through oracle/_ref (tests/golden/lutf.npz); the other functions are unpinned
 * (their translation units need glibmm/lcms2 headers).
 */
#include "oracle.h"
#include "oracle_common.h"

void oracle_get_image(const float *const src[3], size_t ss, int sx1, int sy1,
                      float *const dst[3], size_t ds, int w, int h, const float mul[3], int do_clip)
{
#pragma omp parallel for
    for (int y = 0; y < h; ++y)
        for (int c = 0; c < 3; ++c) {
            const float *s = src[c] + (size_t)(sy1 + y) * ss + sx1;
            float *d = dst[c] + (size_t)y * ds;
            for (int x = 0; x < w; ++x) {
                float t = 0.f;
                t += s[x];          /* skip == 1: one term of the rtot accumulation (L950-956) */
                t *= mul[c];
                if (do_clip) t = rt_maxf(0.f, rt_minf(t, 65535.f)); /* CLIP = LIM(a,0,MAXVAL) */
                d[x] = t;
            }
        }
}

void oracle_convert_color_space(float *const img[3], size_t s, int w, int h, const double mat[9])
{
#pragma omp parallel for
    for (int y = 0; y < h; ++y) {
        float *r = img[0] + (size_t)y * s, *g = img[1] + (size_t)y * s, *b = img[2] + (size_t)y * s;
        for (int x = 0; x < w; ++x) {
            float nr = mat[0] * r[x] + mat[1] * g[x] + mat[2] * b[x];
            float ng = mat[3] * r[x] + mat[4] * g[x] + mat[5] * b[x];
            float nb = mat[6] * r[x] + mat[7] * g[x] + mat[8] * b[x];
            r[x] = nr; g[x] = ng; b[x] = nb;
        }
    }
}

void oracle_exposure(float *const img[3], size_t s, int w, int h, float exp_scale, float black)
{
    const int wv = (w / 4) * 4; /* SSE loop `for (; x < W - 3; x += 4)` covers [0, 4*floor(W/4)) */
#pragma omp parallel for
    for (int y = 0; y < h; ++y)
        for (int c = 0; c < 3; ++c) {
            float *p = img[c] + (size_t)y * s;
            for (int x = 0; x < w; ++x) {
                float t = p[x] * exp_scale - black;
                p[x] = x < wv ? sse_maxf(t, 0.f) : std_maxf(t, 0.f);
            }
        }
}

static inline void clip_rgb_tone(float *r, float *g, float *b, float L)
{
    float r_ = *r > L ? L : *r;
    float b_ = *b > L ? L : *b;
    float g_ = b_ + ((r_ - b_) * (*g - *b) / (*r - *b));
    *r = r_; *g = g_; *b = b_;
}

static inline void filmlike_clip_px(float *r, float *g, float *b, float L)
{
    if (*r >= *g) {
        if (*g > *b) clip_rgb_tone(r, g, b, L);
        else if (*b > *r) clip_rgb_tone(b, r, g, L);
        else if (*b > *g) clip_rgb_tone(r, b, g, L);
        else { *r = *r > L ? L : *r; *g = *g > L ? L : *g; *b = *g; }
    } else {
        if (*r >= *b) clip_rgb_tone(g, r, b, L);
        else if (*b > *g) clip_rgb_tone(b, g, r, L);
        else clip_rgb_tone(g, b, r, L);
    }
}

void oracle_filmlike_clip(float *const img[3], size_t s, int w, int h, float whitept)
{
    const float Lmax = 65535.f * whitept;
#pragma omp parallel for
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            size_t o = (size_t)y * s + x;
            filmlike_clip_px(&img[0][o], &img[1][o], &img[2][o], Lmax);
        }
}

/* LUTf::operator[](float) with the default LUT_CLIP_BELOW|LUT_CLIP_ABOVE (LUT.h:436-459) */
float oracle_lutf(const float *data, int size, float index)
{
    const int maxs = size - 2, upper = size - 1;
    int idx = (int)index;
    if (index < 0.f || !(index == index)) return data[0];
    if (index > (float)maxs) return data[upper];
    float diff = index - (float)idx;
    float p1 = data[idx];
    float p2 = data[idx + 1] - p1;
    return p1 + p2 * diff;
}

/* curves::setLutVal (curves.h:224-231): val <= 65535 or no Curve object -> lut[max(val, 0)]; above, curve->getVal(val / 65535.f) *
   65535.f.  What getVal returns above 1.0 depends on the curve kind (diagonalcurves.cc:443-561): the last point's y for DCT_Linear /
   DCT_Spline / DCT_CatmullRom (L476-477, L514-515), t itself for DCT_Empty and DCT_NURBS beyond its hash table (L529-535, L557-560);
   DCT_Parametric continues analytically (kind 4, below).  oracle_curve_tail_kind: 0 no Curve object, 1 constant, 2 identity, 4 parametric. */
int oracle_curve_tail_kind = 0;
double oracle_curve_tail_y = 1.0;

/* DCT_Parametric (kind 4): the elementary curves of curves.h:92-156, DiagonalCurve's constructor (diagonalcurves.cc:106-131) and
   getVal (L448-470), in double with the reference's sleef xlog / xexp (oracle/sleef.c; pinned by tests/golden/sleef_d.npz) */
double oracle_xlog(double d);
double oracle_xexp(double d);
static double pc_basel(double x, double m1, double m2)
{
    if (x == 0.0) return 0.0;
    double k = sqrt((m1 - 1.0) * (m1 - m2) * 0.5) / (1.0 - m2);
    double l = (m1 - m2) / (1.0 - m2) + k;
    double lx = oracle_xlog(x);
    return m2 * x + (1.0 - m2) * (2.0 - oracle_xexp(k * lx)) * oracle_xexp(l * lx);
}
static double pc_baseu(double x, double m1, double m2) { return 1.0 - pc_basel(1.0 - x, m1, m2); }
static double pc_cupper(double x, double m, double hr)
{
    if (hr > 1.0) return pc_baseu(x, m, 2.0 * (hr - 1.0) / m);
    double x1 = (1.0 - hr) / m;
    double x2 = x1 + hr;
    if (x >= x2) return 1.0;
    if (x < x1) return x * m;
    return 1.0 - hr + hr * pc_baseu((x - x1) / hr, m, 0);
}
static double pc_clower(double x, double m, double sr) { return 1.0 - pc_cupper(1.0 - x, m, sr); }
static double pc_p00(double x, double prot) { return pc_clower(x, 2.0, prot); }
static double pc_p11(double x, double prot) { return pc_cupper(x, 2.0, prot); }
static double pc_p01(double x, double prot) { return x <= 0.5 ? pc_clower(x * 2, 2.0, prot) * 0.5 : 0.5 + pc_cupper((x - 0.5) * 2, 2.0, prot) * 0.5; }
static double pc_p10(double x, double prot) { return x <= 0.5 ? pc_cupper(x * 2, 2.0, prot) * 0.5 : 0.5 + pc_clower((x - 0.5) * 2, 2.0, prot) * 0.5; }
static double pc_pfull(double x, double prot, double sh, double hl)
{
    return (1 - sh) * (1 - hl) * pc_p00(x, prot) + sh * hl * pc_p11(x, prot) + (1 - sh) * hl * pc_p01(x, prot) + sh * (1 - hl) * pc_p10(x, prot);
}
static double pcx[9], pc_mc, pc_mfc, pc_msc, pc_mhc;
void oracle_set_parametric_curve(const double *p, int np)
{
    pcx[0] = p[0];
    for (int i = 1; i < 4; i++) { double v = p[i] > 0.001 ? p[i] : 0.001; pcx[i] = v < 0.99 ? v : 0.99; }
    for (int i = 4; i < 8; i++) pcx[i] = (p[i] + 100.0) / 200.0;
    pcx[8] = np < 9 ? 1.0 : p[8] / 100.0;
    pc_mc = -oracle_xlog(2.0) / oracle_xlog(pcx[2]);
    double mbase = pc_pfull(0.5, pcx[8], pcx[6], pcx[5]);
    pc_mfc = mbase <= 1e-14 ? 0.0 : oracle_xexp(oracle_xlog(mbase) / pc_mc);
    pc_msc = -oracle_xlog(2.0) / oracle_xlog(pcx[1] / pcx[2]);
    pc_mhc = -oracle_xlog(2.0) / oracle_xlog((pcx[3] - pcx[2]) / (1 - pcx[2]));
    oracle_curve_tail_kind = 4;
}
double oracle_parametric_getval(double t)
{
    if (t <= 1e-14) return 0.0;
    double tv = oracle_xexp(pc_mc * oracle_xlog(t));
    double base = pc_pfull(tv, pcx[8], pcx[6], pcx[5]);
    double stretched = base <= 1e-14 ? 0.0 : oracle_xexp(oracle_xlog(base) / pc_mc);
    if (t < pcx[2]) {
        double stv = oracle_xexp(pc_msc * oracle_xlog(stretched / pc_mfc));
        double sbase = pc_pfull(stv, pcx[8], pcx[7], 0.5);
        return pc_mfc * (sbase <= 1e-14 ? 0.0 : oracle_xexp(oracle_xlog(sbase) / pc_msc));
    }
    double htv = oracle_xexp(pc_mhc * oracle_xlog((stretched - pc_mfc) / (1 - pc_mfc)));
    double hbase = pc_pfull(htv, pcx[8], 0.5, pcx[4]);
    return pc_mfc + (1 - pc_mfc) * (hbase <= 1e-14 ? 0.0 : oracle_xexp(oracle_xlog(hbase) / pc_mhc));
}
void oracle_t_parametric_getval(const double *t, double *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = oracle_parametric_getval(t[i]); }

float oracle_set_lut_val(const float *lut65536, float val)
{
    if (val <= 65535.f || oracle_curve_tail_kind == 0) return oracle_lutf(lut65536, 65536, std_maxf(val, 0.f));
    const double t = (double)(val / 65535.f);
    if (oracle_curve_tail_kind == 4) return (float)(oracle_parametric_getval(t) * (double)65535.f);
    return (float)((oracle_curve_tail_kind == 1 ? oracle_curve_tail_y : t) * (double)65535.f);
}

void oracle_tone_curve_std(float *const img[3], size_t s, int w, int h, const float *lut65536)
{
#pragma omp parallel for
    for (int y = 0; y < h; ++y)
        for (int c = 0; c < 3; ++c) {
            float *p = img[c] + (size_t)y * s;
            for (int x = 0; x < w; ++x) {
                p[x] = oracle_set_lut_val(lut65536, p[x]);
            }
        }
}

/* Imagefloat::setMode(YUV) / setMode(RGB) (imagefloat.cc:700-725,779-804), float working-space matrix */
void oracle_rgb_to_yuv(float *const img[3], size_t s, int w, int h, const float ws[9])
{
#pragma omp parallel for
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            size_t o = (size_t)y * s + x;
            float r = img[0][o], g = img[1][o], b = img[2][o];
            float Y = r * ws[3] + g * ws[4] + b * ws[5];
            img[1][o] = Y; img[2][o] = Y - b; img[0][o] = r - Y;
        }
}
void oracle_yuv_to_rgb(float *const img[3], size_t s, int w, int h, const float ws[9])
{
#pragma omp parallel for
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            size_t o = (size_t)y * s + x;
            float Y = img[1][o], u = img[2][o], v = img[0][o];
            float b = Y - u, r = v + Y;
            float g = (Y - r * ws[3] - b * ws[5]) / ws[4];
            img[0][o] = r; img[1][o] = g; img[2][o] = b;
        }
}

/* copyOriginalPixels (no dark frame / flat field) + scaleColors (rawimagesource.cc:2325-2428, 2739-2760, 2806-2813).
 * cfa36: colour at [row % 6][col % 6]; bayer: the greens of even rows use index 3. */
void oracle_scale_colors(const void *src, int src_u16, int w, int h, const int cfa36[36], int bayer, const float cblacksom[4],
                         const float scale_mul[4], float *dst, float chmax[4])
{
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
#pragma omp parallel for reduction(max : m0, m1, m2)
    for (int row = 0; row < h; ++row)
        for (int col = 0; col < w; ++col) {
            const size_t i = (size_t)row * w + col;
            float val = src_u16 ? (float)((const unsigned short *)src)[i] : ((const float *)src)[i];
            const int c = cfa36[(row % 6) * 6 + col % 6];
            const int c4 = (bayer && c == 1 && !(row & 1)) ? 3 : c;
            val = rt_maxf(0.f, val - cblacksom[c4]) * scale_mul[c4];
            dst[i] = val;
            if (c == 0) m0 = rt_maxf(m0, val); else if (c == 1) m1 = rt_maxf(m1, val); else m2 = rt_maxf(m2, val);
        }
    chmax[0] = m0; chmax[1] = m1; chmax[2] = m2; chmax[3] = m1;
}

/* channelMixer pixel loop (ipchmixer.cc:200-230) */
void oracle_channel_mixer(float *const img[3], size_t s, int w, int h, const float m[9])
{
#pragma omp parallel for
    for (int y = 0; y < h; ++y) {
        int x = 0;
        for (; x < w - 3; x += 4)
            for (int k = 0; k < 4; ++k) {
                const size_t o = (size_t)y * s + x + k;
                const float r = img[0][o], g = img[1][o], b = img[2][o];
                img[0][o] = sse_maxf((r * m[0] + g * m[1] + b * m[2]), 0.f);
                img[1][o] = sse_maxf((r * m[3] + g * m[4] + b * m[5]), 0.f);
                img[2][o] = sse_maxf((r * m[6] + g * m[7] + b * m[8]), 0.f);
            }
        for (; x < w; ++x) {
            const size_t o = (size_t)y * s + x;
            const float r = img[0][o], g = img[1][o], b = img[2][o];
            img[0][o] = rt_maxf((r * m[0] + g * m[1] + b * m[2]), 0.f);
            img[1][o] = rt_maxf((r * m[3] + g * m[4] + b * m[5]), 0.f);
            img[2][o] = rt_maxf((r * m[6] + g * m[7] + b * m[8]), 0.f);
        }
    }
}
/* rgbCurves pixel loop (iprgbcurves.cc:116-143); luts[c] may be NULL */
void oracle_rgb_curves(float *const img[3], size_t s, int w, int h, const float *const luts[3])
{
#pragma omp parallel for
    for (int y = 0; y < h; ++y)
        for (int c = 0; c < 3; ++c) {
            if (!luts[c]) continue;
            int x = 0;
            for (; x < w - 3; x += 4)
                for (int k = 0; k < 4; ++k) img[c][(size_t)y * s + x + k] = oracle_lutf_vec(luts[c], 65536, img[c][(size_t)y * s + x + k]);
            for (; x < w; ++x) img[c][(size_t)y * s + x] = oracle_lutf_noclip(luts[c], 65536, img[c][(size_t)y * s + x]);
        }
}

/* getImage with PreviewProps::skip > 1 (rawimagesource.cc:940-975): skip x skip sums, rows outer; W, H = plane size */
void oracle_get_image_skip(const float *const src[3], size_t ss, int W, int H, int sx1, int sy1, int skip,
                           float *const dst[3], size_t ds, int w, int h, const float mul[3], int do_clip)
{
#pragma omp parallel for
    for (int ix = 0; ix < h; ++ix) {
        int i = sy1 + skip * ix;
        i = i < H - skip ? i : H - skip;
        for (int j = 0, jx = sx1; j < w; j++, jx += skip) {
            jx = jx < W - skip ? jx : W - skip;
            float tot[3] = {0.f, 0.f, 0.f};
            for (int m = 0; m < skip; m++)
                for (int n = 0; n < skip; n++)
                    for (int c = 0; c < 3; ++c) tot[c] += src[c][(size_t)(i + m) * ss + jx + n];
            for (int c = 0; c < 3; ++c) {
                float t = tot[c] * mul[c];
                if (do_clip) t = rt_maxf(0.f, rt_minf(t, 65535.f));
                dst[c][(size_t)ix * ds + j] = t;
            }
        }
    }
}

/* ImProcFunctions::saturationVibrance (ipsaturation.cc:29-83) */
static float apply_vibrance(float x, float vib, float noise)
{
    const float ax = fabsf(x / 65535.f);
    if (ax > noise) return (float)((0.f < x) - (x < 0.f)) * oracle_pow_F(ax, vib) * 65535.f;
    return x;
}
void oracle_saturation_vibrance(float *const img[3], size_t s, int w, int h, int saturation_p, int vibrance_p, const double ws[9])
{
    const float saturation = 1.f + saturation_p / 100.f, vibrance = 1.f - vibrance_p / 1000.f;
    const float noise = oracle_pow_F(2.f, -16.f);
    if (!saturation_p && !vibrance_p) return;
#pragma omp parallel for
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) {
            const size_t o = (size_t)i * s + j;
            const float r = img[0][o], g = img[1][o], b = img[2][o];
            const float l = r * ws[3] + g * ws[4] + b * ws[5];
            float rl = r - l, gl = g - l, bl = b - l;
            if (vibrance_p) { rl = apply_vibrance(rl, vibrance, noise); gl = apply_vibrance(gl, vibrance, noise); bl = apply_vibrance(bl, vibrance, noise); }
            img[0][o] = rt_maxf(l + saturation * rl, noise);
            img[1][o] = rt_maxf(l + saturation * gl, noise);
            img[2][o] = rt_maxf(l + saturation * bl, noise);
        }
}

/* ARTOutputProfile::operator()(const Imagefloat*, Imagefloat*) (iprgb2out.cc:152-172): matrix (float, accumulated from 0 in column
 * order: linalgebra.h:227-239) + TRC from a LUT for values <= 1; linear mode passes values through.  Returns the number of channel
 * values that would need ARTOutputProfile::eval (lcms2 / libm); they are left as the matrix output. */
int oracle_rgb2out_matrix(const float *const src[3], float *const dst[3], size_t s, int w, int h, const float m[9], int linear,
                          const float *lut, int lutsz)
{
    const float factor = (float)(lutsz - 1);
    int bad = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t o = (size_t)y * s + x;
            const float rgb[3] = {src[0][o] / 65535.f, src[1][o] / 65535.f, src[2][o] / 65535.f};
            for (int i = 0; i < 3; ++i) {
                float acc = 0;
                for (int k = 0; k < 3; ++k) acc += m[3 * i + k] * rgb[k];
                if (lutsz > 0 && acc <= 1.f) acc = oracle_lutf(lut, lutsz, acc * factor);
                else if (!linear) ++bad;
                dst[i][o] = acc * 65535.f;
            }
        }
    return bad;
}
/* DNG_FloatToHalf (halffloat.h:9-46) */
static unsigned short float_to_half_dng(float f)
{
    union { float f; uint32_t i; } tmp;
    tmp.f = f;
    int32_t sign = (tmp.i >> 16) & 0x00008000;
    int32_t exponent = ((tmp.i >> 23) & 0x000000ff) - (127 - 15);
    int32_t mantissa = tmp.i & 0x007fffff;
    if (exponent <= 0) {
        if (exponent < -10) return (unsigned short)sign;
        mantissa = (mantissa | 0x00800000) >> (1 - exponent);
        if (mantissa & 0x00001000) mantissa += 0x00002000;
        return (unsigned short)(sign | (mantissa >> 13));
    } else if (exponent == 0xff - (127 - 15)) {
        if (mantissa == 0) return (unsigned short)(sign | 0x7c00);
        return (unsigned short)(sign | 0x7c00 | (mantissa >> 13));
    }
    if (mantissa & 0x00001000) {
        mantissa += 0x00002000;
        if (mantissa & 0x00800000) { mantissa = 0; exponent += 1; }
    }
    if (exponent > 30) return (unsigned short)(sign | 0x7c00);
    return (unsigned short)(sign | (exponent << 10) | (mantissa >> 13));
}
void oracle_float_to_half(const float *x, unsigned short *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = float_to_half_dng(x[i]); }
/* Imagefloat::getScanline for every row (imagefloat.cc:125-170); out: h rows of w*3 samples */
void oracle_get_scanlines(const float *const img[3], size_t s, int w, int h, int bps, int is_float, void *out)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < 3; ++c) {
                const float v = img[c][(size_t)y * s + x];
                const size_t k = ((size_t)y * w + x) * 3 + c;
                if (is_float) {
                    if (bps == 32) ((float *)out)[k] = v / 65535.f;
                    else ((unsigned short *)out)[k] = float_to_half_dng(v / 65535.f);
                } else {
                    const unsigned short q = (unsigned short)rt_maxf(0.f, rt_minf(v, 65535.f));
                    if (bps == 16) ((unsigned short *)out)[k] = q;
                    else ((unsigned char *)out)[k] = (unsigned char)(((q + 128) - ((q + 128) >> 8)) >> 8);
                }
            }
}
