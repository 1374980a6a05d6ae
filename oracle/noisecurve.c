/*
 * oracle/noisecurve.c -- CPU oracle for the chroma noise-curve map of ImProcFunctions::denoise / RGB_denoise.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * Restates
 *   FlatCurve (FCT_MinMaxCPoints)      rtengine/flatcurves.cc:27-77,124-327 (CtrlPoints_set), 329-360 (getVal)
 *   Curve::AddPolygons / fillDyByDx    rtengine/curves.cc:98-132
 *   NoiseCurve::Set                    rtengine/ipdenoise.cc:684-716  (501 samples, floor 0.01, float running sum)
 *   Color::init cachef                 rtengine/color.cc:178,202-217   (65536-entry LUT, LUT_CLIP_BELOW)
 *   Color::computeXYZ2Lab, XYZ2Lab     rtengine/color.cc:1247-1259,1382-1397
 *   xcbrtf                             rtengine/sleef.h:966-991  (pinned against oracle/_ref, tests/golden/sleef.npz)
 *   calclum + ccalc map                rtengine/ipdenoise.cc:1113-1131, rtengine/FTblockDN.cc:1707-1777
 *
 * Parity: the curve classes need glibmm to compile (curves.h:26), so FlatCurve / NoiseCurve are PARITY UNPINNED
 * (restated from the source only); xcbrtf and the LUTf lookup are pinned.
 */
#include "oracle.h"
#include "oracle_common.h"
#include <stdlib.h>

/* ---------------------------------------------------------------- FlatCurve */
typedef struct { double *x, *y; int n, cap; } poly_t;
static void poly_push(poly_t *p, double x, double y)
{
    if (p->n == p->cap) {
        p->cap = p->cap ? 2 * p->cap : 256;
        p->x = (double *)realloc(p->x, sizeof(double) * p->cap);
        p->y = (double *)realloc(p->y, sizeof(double) * p->cap);
    }
    p->x[p->n] = x; p->y[p->n] = y; ++p->n;
}

/* curves.cc:98-120 with firstPointIncluded == false (flatcurves.cc:286) */
static void add_polygons(poly_t *p, int nbr_points, double increment, double x1, double y1, double x2, double y2, double x3, double y3)
{
    (void)x1; (void)y1;
    for (int k = 1; k < nbr_points - 1; k++) {
        double t = k * increment;
        double t2 = t * t;
        double tr = 1. - t;
        double tr2 = tr * tr;
        double tr2t = tr * 2 * t;
        poly_push(p, tr2 * x1 + tr2t * x2 + t2 * x3, tr2 * y1 + tr2t * y2 + t2 * y3);
    }
    poly_push(p, x3, y3);
}

/* Samples FlatCurve(points, periodic, ppn) after setIdentityValue(identity) at t = i/(nout-1), i = 0..nout-1.
 * points: {kind, x0, y0, lt0, rt0, x1, ...}.  Returns 1 when the curve is the identity (kind FCT_Empty), else 0. */
/* FlatCurve's constructor: builds the polyline getVal works on; returns 1 for an identity / empty curve (then *pout is empty) */
static int flat_curve_build(const double *pts, int npts, int periodic, int ppn_in, double identity, poly_t *pout, double **dydx_out)
{
    pout->x = pout->y = NULL; pout->n = pout->cap = 0; *dydx_out = NULL;
    const int FCT_MinMaxCPoints = 1;
    int is_identity = 1;
    if (npts > 4 && (int)pts[0] == FCT_MinMaxCPoints) {
        const int one_more = periodic ? 1 : 0;
        const int N = (npts - 1) / 4;
        double *x = (double *)malloc(sizeof(double) * (N + 1)), *y = (double *)malloc(sizeof(double) * (N + 1));
        double *lt = (double *)malloc(sizeof(double) * (N + 1)), *rt = (double *)malloc(sizeof(double) * (N + 1));
        for (int i = 0, ix = 1; i < N; i++) { x[i] = pts[ix++]; y[i] = pts[ix++]; lt[i] = pts[ix++]; rt[i] = pts[ix++]; }
        if (periodic) { x[N] = pts[1] + 1.0; y[N] = pts[2]; lt[N] = pts[3]; rt[N] = pts[4]; }
        is_identity = 1;
        for (int i = 0; i < N + one_more; i++)
            if (y[i] >= identity + 1.e-7 || y[i] <= identity - 1.e-7) { is_identity = 0; break; }
        if (!is_identity && N > one_more) {
            const int ppn = ppn_in > 65500 ? 65500 : ppn_in;
            const int N_ = periodic ? N : N - 1;
            double *sc_x = (double *)calloc((size_t)N_ * 6, sizeof(double)), *sc_y = (double *)calloc((size_t)N_ * 6, sizeof(double));
            double *sc_len = (double *)calloc((size_t)N_ * 2, sizeof(double));
            int *sc_lin = (int *)calloc((size_t)N_ * 2, sizeof(int));
            double total_length = 0.;
            unsigned j = 0, k = 0;
#define SEG_LEN() sqrt((sc_x[j] - sc_x[j - 1]) * (sc_x[j] - sc_x[j - 1]) + (sc_y[j] - sc_y[j - 1]) * (sc_y[j] - sc_y[j - 1]))
            for (int i = 0; i < N_;) {
                double length;
                const int start_linear = (rt[i] == 0.) || (y[i] == y[i + 1]);
                const int end_linear = (lt[i + 1] == 0.) || (y[i] == y[i + 1]);
                if (start_linear && end_linear) {
                    sc_x[j] = x[i]; sc_y[j++] = y[i];
                    sc_x[j] = x[i + 1]; sc_y[j] = y[i + 1];
                    sc_lin[k] = 1;
                    i++;
                    length = SEG_LEN();
                    j++;
                    sc_len[k++] = length; total_length += length;
                } else {
                    double xp1 = start_linear ? x[i] : (x[i + 1] - x[i]) * rt[i] + x[i];
                    double xp3 = end_linear ? x[i + 1] : (x[i] - x[i + 1]) * lt[i + 1] + x[i + 1];
                    const double xp2 = (xp1 + xp3) / 2.0, yp2 = (y[i] + y[i + 1]) / 2.0;
                    if (rt[i] + lt[i + 1] > 1.0) xp1 = xp3 = xp2;
                    if (start_linear) {
                        sc_x[j] = x[i]; sc_y[j++] = y[i];
                        sc_x[j] = xp2; sc_y[j] = yp2;
                        sc_lin[k] = 1;
                        length = SEG_LEN();
                        j++;
                        sc_len[k++] = length; total_length += length;
                    } else {
                        sc_x[j] = x[i]; sc_y[j++] = y[i];
                        sc_x[j] = xp1; sc_y[j] = y[i];
                        length = SEG_LEN();
                        j++;
                        sc_x[j] = xp2; sc_y[j] = yp2;
                        sc_lin[k] = 0;
                        length += SEG_LEN();
                        j++;
                        sc_len[k++] = length; total_length += length;
                    }
                    if (end_linear) {
                        sc_x[j] = xp2; sc_y[j++] = yp2;
                        sc_x[j] = x[i + 1]; sc_y[j] = y[i + 1];
                        sc_lin[k] = 1;
                        length = SEG_LEN();
                        j++;
                        sc_len[k++] = length; total_length += length;
                    } else {
                        sc_x[j] = xp2; sc_y[j++] = yp2;
                        sc_x[j] = xp3; sc_y[j] = y[i + 1];
                        length = SEG_LEN();
                        j++;
                        sc_x[j] = x[i + 1]; sc_y[j] = y[i + 1];
                        sc_lin[k] = 0;
                        length += SEG_LEN();
                        j++;
                        sc_len[k++] = length; total_length += length;
                    }
                    i++;
                }
            }
#undef SEG_LEN
            poly_t p = {0, 0, 0, 0};
            j = 0;
            if (!periodic && sc_x[j] != 0.) poly_push(&p, 0., sc_y[j]);
            poly_push(&p, sc_x[j], sc_y[j]);
            for (unsigned i = 0; i < k; i++) {
                if (sc_lin[i]) {
                    j++;
                    poly_push(&p, sc_x[j], sc_y[j]);
                    j++;
                } else {
                    const int nbr_points = (int)(((double)ppn * sc_len[i]) / total_length);
                    const double increment = 1.0 / (double)(nbr_points - 1);
                    const double x1 = sc_x[j], y1 = sc_y[j]; j++;
                    const double x2 = sc_x[j], y2 = sc_y[j]; j++;
                    const double x3 = sc_x[j], y3 = sc_y[j]; j++;
                    add_polygons(&p, nbr_points, increment, x1, y1, x2, y2, x3, y3);
                }
            }
            poly_push(&p, 3.0, sc_y[j - 1]);
            double *dy_by_dx = (double *)malloc(sizeof(double) * (p.n - 1));
            for (int i = 0; i < p.n - 1; i++) dy_by_dx[i] = (p.y[i + 1] - p.y[i]) / (p.x[i + 1] - p.x[i]);
            *pout = p; *dydx_out = dy_by_dx;
            free(sc_x); free(sc_y); free(sc_len); free(sc_lin);
        } else {
            is_identity = 1;
        }
        free(x); free(y); free(lt); free(rt);
    }
    return is_identity;
}



/* FlatCurve::getVal, FCT_MinMaxCPoints (flatcurves.cc:344-365) */
static double flat_curve_val(const poly_t *p, const double *dydx, double t)
{
    if (t < p->x[0]) t += 1.0;
    unsigned k_lo = 0, k_hi = (unsigned)p->n - 1;
    while (k_hi > 1 + k_lo) {
        const unsigned m = (k_hi + k_lo) / 2;
        if (p->x[m] > t) k_hi = m; else k_lo = m;
    }
    return p->y[k_lo] + (t - p->x[k_lo]) * dydx[k_lo];
}
int oracle_flat_curve_sample(const double *pts, int npts, int periodic, int ppn_in, double identity, int nout, double *out)
{
    poly_t p; double *dydx;
    const int is_identity = flat_curve_build(pts, npts, periodic, ppn_in, identity, &p, &dydx);
    for (int s = 0; s < nout; ++s) out[s] = is_identity ? identity : flat_curve_val(&p, dydx, (double)s / (double)(nout - 1));
    free(dydx); free(p.x); free(p.y);
    return is_identity;
}
/* handle form for per-pixel evaluation (hslEqualizer) */
typedef struct { poly_t p; double *dydx; int identity; double identity_value; } oracle_flat_curve;
void *oracle_flat_curve_new(const double *pts, int npts, int periodic, int ppn, double identity)
{
    oracle_flat_curve *c = (oracle_flat_curve *)calloc(1, sizeof *c);
    c->identity_value = identity;
    c->identity = (pts == NULL) ? 1 : flat_curve_build(pts, npts, periodic, ppn, identity, &c->p, &c->dydx);
    return c;
}
int oracle_flat_curve_is_identity(const void *h) { return ((const oracle_flat_curve *)h)->identity; }
double oracle_flat_curve_get(const void *h, double t)
{
    const oracle_flat_curve *c = (const oracle_flat_curve *)h;
    return c->identity ? c->identity_value : flat_curve_val(&c->p, c->dydx, t);
}
void oracle_flat_curve_free(void *h)
{
    oracle_flat_curve *c = (oracle_flat_curve *)h;
    free(c->dydx); free(c->p.x); free(c->p.y); free(c);
}

/* NoiseCurve::Set(const std::vector<double>&) (ipdenoise.cc:705-716 -> 684-703): non-periodic FlatCurve with
 * ppn = CURVES_MIN_POLY_POINTS/2 = 500, identity value 0.  Returns the float sum (0 and an all-zero LUT if reset). */
float oracle_noise_curve(const double *pts, int npts, float lut[501])
{
    double v[501];
    for (int i = 0; i < 501; ++i) lut[i] = 0.f;
    if (!(npts > 0 && pts[0] > 0 /*FCT_Linear*/ && pts[0] < 2 /*FCT_Unchanged*/)) return 0.f;
    if (oracle_flat_curve_sample(pts, npts, 0, 500, 0., 501, v)) return 0.f;
    float sum = 0.f;
    for (int i = 0; i < 501; ++i) {
        lut[i] = (float)v[i];
        if (lut[i] < 0.01f) lut[i] = 0.01f;
        sum += lut[i];
    }
    return sum;
}

/* ---------------------------------------------------------------- Lab LUT + chroma map */
#define MAXVALF 65535.f
static const double KAPPA = 24389.0 / 27.0;
static const double EPS_LAB = 216.0 / 24389.0;

void oracle_cachef(float lut[65536])
{
    const double eps_max = (double)MAXVALF * EPS_LAB;
    int i = 0;
    const int epsmaxint = (int)eps_max;
    for (; i <= epsmaxint; i++) lut[i] = (float)(327.68 * ((KAPPA * i / MAXVALF + 16.0) / 116.0));
    for (; i < 65536; i++) lut[i] = (float)(327.68 * cbrt((double)i / MAXVALF));
}

float oracle_xyz2lab_f(const float *cachef, float f);
static float xyz2lab_f(const float *cachef, float f) { return oracle_xyz2lab_f(cachef, f); }
float oracle_xyz2lab_f(const float *cachef, float f)
{
    if (f != f) return f;
    if (f < 0.f) return (float)(327.68 * ((KAPPA * f / MAXVALF + 16.0) / 116.0));
    if (f > 65535.f) return 327.68f * oracle_xcbrtf(f / MAXVALF);
    /* LUT_CLIP_BELOW only (color.cc:178) */
    const int maxs = 65534;
    int idx = (int)f;
    if (f > (float)maxs) idx = maxs;
    const float diff = f - (float)idx;
    const float p1 = cachef[idx], p2 = cachef[idx + 1] - p1;
    return p1 + p2 * diff;
}

/* ipdenoise.cc:1113-1131 (calclum = every second pixel, then convertColorSpace's matrix branch with double
 * accumulation, rawimagesource.cc:3184-3213) + FTblockDN.cc:1716-1777 (useNoiseCCurve only; the luminance curve is
 * never set in ART).  out is ((w+1)/2) x ((h+1)/2).  mat may be NULL (no conversion). */
void oracle_chroma_noise_map(const float *const img[3], size_t s, int w, int h, const double *mat, const float wpi[9],
                             const float curve[501], float *out)
{
    static float *cachef = NULL;
    if (!cachef) { cachef = (float *)malloc(sizeof(float) * 65536); oracle_cachef(cachef); }
    const int wid = (w + 1) / 2, hei = (h + 1) / 2;
    const float D50x = 0.9642f, D50z = 0.8249f;
    const float t0 = 1.f + 1.f * (4.f * oracle_lutf(curve, 501, 100.f / 60.f));
    const float cn100 = t0 * t0;
#pragma omp parallel for
    for (int ii = 0; ii < hei; ++ii)
        for (int jj = 0; jj < wid; ++jj) {
            const size_t o = (size_t)(2 * ii) * s + 2 * jj;
            float RL = img[0][o], GL = img[1][o], BL = img[2][o];
            if (mat) {
                const double dr = RL, dg = GL, db = BL;
                RL = (float)(mat[0] * dr + mat[1] * dg + mat[2] * db);
                GL = (float)(mat[3] * dr + mat[4] * dg + mat[5] * db);
                BL = (float)(mat[6] * dr + mat[7] * dg + mat[8] * db);
            }
            const float XL = wpi[0] * RL + wpi[1] * GL + wpi[2] * BL;
            const float YL = wpi[3] * RL + wpi[4] * GL + wpi[5] * BL;
            const float ZL = wpi[6] * RL + wpi[7] * GL + wpi[8] * BL;
            const float fx = xyz2lab_f(cachef, XL / D50x), fy = xyz2lab_f(cachef, YL), fz = xyz2lab_f(cachef, ZL / D50z);
            const float A = 500.0f * (fx - fy), B = 200.0f * (fy - fz);
            const float cN = sqrtf(A * A + B * B);
            float r;
            if (cN > 100) {
                const float t = 1.f + 1.f * (4.f * oracle_lutf(curve, 501, cN / 60.f));
                r = t * t;
            } else {
                r = cn100;
            }
            out[(size_t)ii * wid + jj] = r;
        }
}

/* ---------------------------------------------------------------- Lab helpers of RGB_denoise's LAB colour-space mode */
/* Color::cachefy (color.cc:219-234), LUT_CLIP_BELOW */
void oracle_cachefy(float lut[65536])
{
    const double eps_max = (double)MAXVALF * EPS_LAB;
    int i = 0;
    const int epsmaxint = (int)eps_max;
    for (; i <= epsmaxint; i++) lut[i] = (float)(327.68 * (KAPPA * i / MAXVALF));
    for (; i < 65536; i++) lut[i] = (float)(327.68 * (116.0 * cbrt((double)i / MAXVALF) - 16.0));
}
/* Color::denoiseGammaTab / denoiseIGammaTab (color.cc:278-292) with gamma55 / igamma55 (color.h:1155-1169) */
void oracle_denoise_gamma_tabs(float *gtab, float *igtab)
{
    for (int i = 0; i < 65536; i++) {
        const double x = i / 65535.0;
        gtab[i] = (float)(65535.0 * (x <= 0.013189 ? x * 10.0 : 1.593503 * exp(log(x) / 5.5) - 0.593503));
        igtab[i] = (float)(65535.0 * (x <= 0.131889 ? x / 10.0 : exp(log((x + 0.593503) / 1.593503) * 5.5)));
    }
}
/* LUTf::operator[](float) of a LUT constructed with flags 0: extrapolates on both sides (LUT.h:436-459) */
float oracle_lutf_noclip(const float *data, int size, float index)
{
    const int maxs = size - 2;
    int idx = (int)index;
    if (index < 0.f || !(index == index)) idx = 0;
    else if (index > (float)maxs) idx = maxs;
    const float diff = index - (float)idx;
    const float p1 = data[idx], p2 = data[idx + 1] - p1;
    return p1 + p2 * diff;
}
static float *g_cf, *g_cfy;
static void lab_luts(void)
{
    if (!g_cf) {
        float *a = (float *)malloc(sizeof(float) * 65536), *b = (float *)malloc(sizeof(float) * 65536);
        oracle_cachef(a); oracle_cachefy(b);
        g_cfy = b; g_cf = a;
    }
}
/* Color::rgb2lab(R,G,B, float ws) = rgbxyz + XYZ2Lab (color.h:630-636, color.cc:1247-1275,1382-1397) */
void oracle_rgb2lab(float R, float G, float B, float *l, float *a, float *b, const float ws[9])
{
    lab_luts();
    const float X = ws[0] * R + ws[1] * G + ws[2] * B, Y = ws[3] * R + ws[4] * G + ws[5] * B, Z = ws[6] * R + ws[7] * G + ws[8] * B;
    const float x = X / 0.9642f, z = Z / 0.8249f, y = Y;
    const float fx = oracle_xyz2lab_f(g_cf, x), fy = oracle_xyz2lab_f(g_cf, y), fz = oracle_xyz2lab_f(g_cf, z);
    float L;
    if (y != y) L = y;
    else if (y < 0.f) L = (float)(327.68 * (KAPPA * y / MAXVALF));
    else if (y > 65535.f) L = 327.68f * (116.f * oracle_xcbrtf(y / MAXVALF) - 16.f);
    else {
        int idx = (int)y;
        if (y > 65534.f) idx = 65534;
        const float diff = y - (float)idx, p1 = g_cfy[idx], p2 = g_cfy[idx + 1] - p1;
        L = p1 + p2 * diff;
    }
    *l = L; *a = 500.0f * (fx - fy); *b = 200.0f * (fy - fz);
}
/* Color::lab2rgb = Lab2XYZ + xyz2rgb (color.h:638-644, color.cc:1203-1214, color.h:767-770) */
void oracle_lab2rgb(float l, float a, float b, float *R, float *G, float *B, const float iws[9])
{
    const float c1By116 = (float)(1.0 / 116.0), c16By116 = (float)(16.0 / 116.0);
    const float epsilonExpInv3f = (float)(6.0 / 29.0), kappaInvf = (float)(27.0 / 24389.0);
    const float LL = l / 327.68f, aa = a / 327.68f, bb = b / 327.68f;
    const float fy = (c1By116 * LL) + c16By116;
    const float fx = (0.002f * aa) + fy;
    const float fz = fy - (0.005f * bb);
#define F2XYZ(f) (((f) > epsilonExpInv3f) ? (f) * (f) * (f) : (116.f * (f) - 16.f) * kappaInvf)
    const float x = 65535.0f * F2XYZ(fx) * 0.9642f;
    const float z = 65535.0f * F2XYZ(fz) * 0.8249f;
#undef F2XYZ
    const float y = (LL > 8.0) ? 65535.0f * fy * fy * fy : (float)(65535.0f * LL / KAPPA);
    *R = iws[0] * x + iws[1] * y + iws[2] * z;
    *G = iws[3] * x + iws[4] * y + iws[5] * z;
    *B = iws[6] * x + iws[7] * y + iws[8] * z;
}
