// hostluts.hip -- host-side construction of the small look-up tables the device stages take as inputs.  These are
// the counterparts of tables the reference also builds once on the host (no per-pixel work here):
//   FlatCurve (FCT_MinMaxCPoints) polyline     rtengine/flatcurves.cc:27-77,124-360, rtengine/curves.cc:98-132
//   NoiseCurve::Set -> 501-entry LUT + sum     rtengine/ipdenoise.cc:684-716
//   Color::cachef                              rtengine/color.cc:178,202-217
#include "kernels.h"
#include <algorithm>
#include <cmath>
#include <vector>

namespace artgpu {

namespace {

struct Polyline {
    std::vector<double> x, y, slope;
    void add(double px, double py) { x.push_back(px); y.push_back(py); }
    // Curve::fillDyByDx
    void finish()
    {
        slope.resize(x.size() - 1);
        for (size_t i = 0; i + 1 < x.size(); ++i) slope[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]);
    }
    // FlatCurve::getVal, FCT_MinMaxCPoints branch
    double eval(double t) const
    {
        if (t < x[0]) t += 1.0;
        unsigned lo = 0, hi = (unsigned)x.size() - 1;
        while (hi > 1 + lo) {
            const unsigned mid = (hi + lo) / 2;
            if (x[mid] > t) hi = mid; else lo = mid;
        }
        return y[lo] + (t - x[lo]) * slope[lo];
    }
};

struct Knot { double x, y, left, right; };

// One sub-curve of FlatCurve::CtrlPoints_set: either a straight segment (2 points) or a quadratic Bezier (3 points).
struct SubCurve { double px[3], py[3]; bool linear; double length; };

double seg(double x0, double y0, double x1, double y1) { const double dx = x1 - x0, dy = y1 - y0; return std::sqrt(dx * dx + dy * dy); }

SubCurve line(double x0, double y0, double x1, double y1)
{
    SubCurve s = {{x0, x1, 0.}, {y0, y1, 0.}, true, 0.};
    s.length = seg(x0, y0, x1, y1);
    return s;
}
SubCurve bezier(double x0, double y0, double x1, double y1, double x2, double y2)
{
    SubCurve s = {{x0, x1, x2}, {y0, y1, y2}, false, 0.};
    s.length = seg(x0, y0, x1, y1);
    s.length += seg(x1, y1, x2, y2);
    return s;
}

// FlatCurve::CtrlPoints_set (flatcurves.cc:124-327)
bool build_polyline(const std::vector<Knot> &k, bool periodic, int ppn, Polyline &poly)
{
    const int nseg = (int)k.size() - 1;   // the caller appended the wrap-around knot for periodic curves
    std::vector<SubCurve> sc;
    double total = 0.;
    for (int i = 0; i < nseg; ++i) {
        const Knot &a = k[i], &b = k[i + 1];
        const bool start_linear = a.right == 0. || a.y == b.y;
        const bool end_linear = b.left == 0. || a.y == b.y;
        if (start_linear && end_linear) {
            sc.push_back(line(a.x, a.y, b.x, b.y));
            total += sc.back().length;
            continue;
        }
        double xp1 = start_linear ? a.x : (b.x - a.x) * a.right + a.x;
        double xp3 = end_linear ? b.x : (a.x - b.x) * b.left + b.x;
        const double xp2 = (xp1 + xp3) / 2.0, yp2 = (a.y + b.y) / 2.0;
        if (a.right + b.left > 1.0) xp1 = xp3 = xp2;
        sc.push_back(start_linear ? line(a.x, a.y, xp2, yp2) : bezier(a.x, a.y, xp1, a.y, xp2, yp2));
        total += sc.back().length;
        sc.push_back(end_linear ? line(xp2, yp2, b.x, b.y) : bezier(xp2, yp2, xp3, b.y, b.x, b.y));
        total += sc.back().length;
    }
    if (sc.empty()) return false;
    if (!periodic && sc[0].px[0] != 0.) poly.add(0., sc[0].py[0]);
    poly.add(sc[0].px[0], sc[0].py[0]);
    double last_y = sc[0].py[0];
    for (const SubCurve &s : sc) {
        if (s.linear) {
            poly.add(s.px[1], s.py[1]);
            last_y = s.py[1];
        } else {
            const int npoints = (int)(((double)ppn * s.length) / total);
            if (npoints < 0) return false;
            const double increment = 1.0 / (double)(npoints - 1);
            for (int q = 1; q < npoints - 1; ++q) {          // Curve::AddPolygons, firstPointIncluded == false
                const double t = q * increment;
                const double t2 = t * t;
                const double tr = 1. - t;
                const double tr2 = tr * tr;
                const double tr2t = tr * 2 * t;
                poly.add(tr2 * s.px[0] + tr2t * s.px[1] + t2 * s.px[2], tr2 * s.py[0] + tr2t * s.py[1] + t2 * s.py[2]);
            }
            poly.add(s.px[2], s.py[2]);
            last_y = s.py[2];
        }
    }
    poly.add(3.0, last_y);
    poly.finish();
    return true;
}

} // namespace

// FlatCurve(points, periodic, ppn): the polyline FlatCurve::getVal works on.  Returns false for an identity / empty curve.
bool flat_curve_polyline(const double *pts, int npts, bool periodic, int ppn, double identity, std::vector<double> &x, std::vector<double> &y, std::vector<double> &slope)
{
    if (!(npts > 4 && (int)pts[0] == 1 /* FCT_MinMaxCPoints */)) return false;
    const int n = (npts - 1) / 4;
    std::vector<Knot> k;
    for (int i = 0; i < n; ++i) k.push_back({pts[1 + 4 * i], pts[2 + 4 * i], pts[3 + 4 * i], pts[4 + 4 * i]});
    if (periodic) k.push_back({pts[1] + 1.0, pts[2], pts[3], pts[4]});
    bool identity_curve = true;
    for (const Knot &q : k)
        if (q.y >= identity + 1.e-7 || q.y <= identity - 1.e-7) { identity_curve = false; break; }
    if (identity_curve || n <= (periodic ? 1 : 0)) return false;
    Polyline poly;
    if (!build_polyline(k, periodic, ppn > 65500 ? 65500 : ppn, poly)) return false;
    x = poly.x; y = poly.y; slope = poly.slope;
    return true;
}

// FlatCurve(points, periodic, ppn) + setIdentityValue(identity), sampled at i/(nout-1).  Returns true if identity.
bool flat_curve_sample(const double *pts, int npts, bool periodic, int ppn, double identity, int nout, double *out)
{
    bool identity_curve = true;
    Polyline poly;
    if (npts > 4 && (int)pts[0] == 1 /* FCT_MinMaxCPoints */) {
        const int n = (npts - 1) / 4;
        std::vector<Knot> k;
        for (int i = 0; i < n; ++i) k.push_back({pts[1 + 4 * i], pts[2 + 4 * i], pts[3 + 4 * i], pts[4 + 4 * i]});
        if (periodic) k.push_back({pts[1] + 1.0, pts[2], pts[3], pts[4]});
        for (const Knot &q : k)
            if (q.y >= identity + 1.e-7 || q.y <= identity - 1.e-7) { identity_curve = false; break; }
        if (!identity_curve && n > (periodic ? 1 : 0)) {
            if (!build_polyline(k, periodic, ppn > 65500 ? 65500 : ppn, poly)) identity_curve = true;
        } else {
            identity_curve = true;
        }
    }
    for (int s = 0; s < nout; ++s) out[s] = identity_curve ? identity : poly.eval((double)s / (double)(nout - 1));
    return identity_curve;
}

// NoiseCurve::Set(const std::vector<double>&): returns the running float sum (0 when the curve is reset)
float noise_curve_lut(const double *pts, int npts, float lut[501])
{
    for (int i = 0; i < 501; ++i) lut[i] = 0.f;
    if (!(npts > 0 && pts[0] > 0. && pts[0] < 2.)) return 0.f;      // FCT_Linear < kind < FCT_Unchanged
    double v[501];
    if (flat_curve_sample(pts, npts, false, 1000 / 2, 0., 501, v)) return 0.f;
    float sum = 0.f;
    for (int i = 0; i < 501; ++i) {
        lut[i] = (float)v[i];
        if (lut[i] < 0.01f) lut[i] = 0.01f;
        sum += lut[i];
    }
    return sum;
}

// Color::cachef
void build_cachef(float *lut)
{
    const double kappa = 24389.0 / 27.0, eps = 216.0 / 24389.0;
    const float maxvalf = 65535.f;
    const int epsmaxint = (int)((double)maxvalf * eps);
    int i = 0;
    for (; i <= epsmaxint; i++) lut[i] = (float)(327.68 * ((kappa * i / maxvalf + 16.0) / 116.0));
    for (; i < 65536; i++) lut[i] = (float)(327.68 * std::cbrt((double)i / maxvalf));
}


// Color::cachefy (color.cc:219-234)
void build_cachefy(float *lut)
{
    const double kappa = 24389.0 / 27.0, eps = 216.0 / 24389.0;
    const float maxvalf = 65535.f;
    const int epsmaxint = (int)((double)maxvalf * eps);
    int i = 0;
    for (; i <= epsmaxint; i++) lut[i] = (float)(327.68 * (kappa * i / maxvalf));
    for (; i < 65536; i++) lut[i] = (float)(327.68 * (116.0 * std::cbrt((double)i / maxvalf) - 16.0));
}

// Color::denoiseGammaTab / denoiseIGammaTab (color.cc:278-292; gamma55 / igamma55, color.h:1155-1169)
void build_denoise_gamma_tabs(float *gtab, float *igtab)
{
    for (int i = 0; i < 65536; i++) {
        const double x = i / 65535.0;
        gtab[i] = (float)(65535.0 * (x <= 0.013189 ? x * 10.0 : 1.593503 * std::exp(std::log(x) / 5.5) - 0.593503));
        igtab[i] = (float)(65535.0 * (x <= 0.131889 ? x / 10.0 : std::exp(std::log((x + 0.593503) / 1.593503) * 5.5)));
    }
}

// Color::init jzazbz_pq_ / jzazbz_pq_inv_ (color.cc:323-326) with PQ / PQ_inv (color.cc:67-86): std::pow(float, float) is the
// host libm's powf, as in the reference
void build_pq_luts(float *pq, float *pq_inv)
{
    for (int i = 0; i < 65536; ++i) {
        const float v = (float)i / 65535.f;
        float X = std::max(v, 1e-10f);
        const float XX = std::pow(X * 1e-4f, 0.1593017578125f);
        pq[i] = std::pow((0.8359375f + 18.8515625f * XX) / (1 + 18.6875f * XX), 134.034375f);
        const float YY = std::pow(X, 7.460772656268214e-03f);
        pq_inv[i] = 1e4f * std::pow((0.8359375f - YY) / (18.6875f * YY - 18.8515625f), 6.277394636015326f);
    }
}

// ---------------------------------------------------------------- AUTOMATIC chrominance: the scalar part of the estimation
// ShrinkAll_info's per-band bookkeeping (FTblockDN.cc:1292-1334) over the 5 levels x 3 directions of one crop.
// mad_a / mad_b: SQR(MadRgb) of the a / b subbands in level-major order (what launch_mad leaves on the device).  out: chaut, maxredaut, maxblueaut, minredaut, minblueaut, Nb
void dninfo_band_stats(const float *mad_a, const float *mad_b, int nbands, bool aggressive, float out[6])
{
    const float reduc = aggressive ? static_cast<float>(0.9) : 1.f;
    float chau = 0.f, maxchred = 0.f, maxchblue = 0.f, minchred = 100000000.f, minchblue = 100000000.f;
    float chaut = 0.f, maxredaut = 0.f, maxblueaut = 0.f, minredaut = 0.f, minblueaut = 0.f;
    int nb = 0;
    for (int k = 0; k < nbands; ++k) {
        const float mada = mad_a[k], madb = mad_b[k];
        maxchred = mada > maxchred ? mada : maxchred;
        minchred = mada < minchred ? mada : minchred;
        maxredaut = std::sqrt(reduc * maxchred);
        minredaut = std::sqrt(reduc * minchred);
        maxchblue = madb > maxchblue ? madb : maxchblue;
        minchblue = madb < minchblue ? madb : minchblue;
        maxblueaut = std::sqrt(reduc * maxchblue);
        minblueaut = std::sqrt(reduc * minchblue);
        chau += (mada + madb);
        ++nb;
        chaut = std::sqrt(reduc * chau / (nb + nb));
    }
    out[0] = chaut; out[1] = maxredaut; out[2] = maxblueaut; out[3] = minredaut; out[4] = minblueaut; out[5] = (float)nb;
}

// calcautodn_info (ipdenoise.cc:66-206) for the arguments its one caller passes: levaut 0, mode 1 ("auto"), lissage 0
// (ipdenoise.cc:893,1003-1004).  Returns delta; chaut is updated in place like the reference's by-reference argument.
static float autodn_adjust(float &chaut, int Nb, float maxmax, float lumema, float chromina, float redyel, float skinc, float nsknc, bool aggressive)
{
    struct Step { float below; float scale; };
    chaut = (chaut * Nb - maxmax) / (Nb - 1);                        // drop the maximum from the mean
    const bool strong_colour = chromina > 3000.f;
    if ((redyel > 5000.f || skinc > 1000.f) && nsknc < 0.4f && strong_colour) chaut *= 0.45f;
    else if ((redyel > 12000.f || skinc > 1200.f) && nsknc < 0.3f && strong_colour) chaut *= 0.3f;
    if (chromina > 10000.f) chaut *= 0.8f;
    else if (chromina > 6000.f) chaut *= 0.9f;
    else if (chromina < 3000.f) chaut *= 1.5f;
    if (lumema < 2500.f) chaut *= 1.2f;
    else if (lumema < 5000.f) chaut *= 1.1f;
    else if (lumema > 20000.f) chaut *= 0.9f;
    if (chaut > 300.f) chaut = 0.714286f * chaut + 85.71428f;      // "low denoise"
    float delta = (maxmax - chaut) * (aggressive ? static_cast<float>(0.9) : 1.f);
    if (chaut < 400.f) {
        const bool low = chaut < 200.f;
        const float knee = low ? 200.f : 400.f;
        if (delta < knee) delta *= low ? 0.95f : 0.6f;
        else if (low && delta < 400.f) delta *= 0.7f;
        else delta = low ? 280.f : 200.f;
    } else {
        static const Step ladder[] = {{550.f, 0.3f}, {650.f, 0.2f}};
        float sc = 0.15f;
        for (const Step &st : ladder) if (chaut < st.below) { sc = st.scale; break; }
        delta *= sc;
    }
    if (chromina < 6000.f) delta *= 1.2f;
    if (lumema < 5000.f) delta *= 1.2f;
    return delta;
}

// The reduction over the nine crops (ipdenoise.cc:960-1072): autoNR 10, autoNRmax 40, multip = adjustr = lowdenoise = 1 (raw).
// info[k] = {chaut, maxredaut, maxblueaut, minredaut, minblueaut, chromina, lumema, redyel, skinc, nsknc, Nb}
void dninfo_reduce(const float info[9][16], bool aggressive, float ch_M[9], float max_r[9], float max_b[9], float out3[3])
{
    const float nrmax = 40.f, nr = 10.f;
    float up_r[9], up_b[9], dn_r[9], dn_b[9];
    for (int k = 0; k < 9; ++k) {
        ch_M[k] = 1.0f * info[k][0]; max_r[k] = 1.0f * info[k][1]; max_b[k] = 1.0f * info[k][2];
        const float min_r = 1.0f * info[k][3], min_b = 1.0f * info[k][4];
        const float delta = autodn_adjust(ch_M[k], (int)info[k][10], std::max(max_r[k], max_b[k]), info[k][6], info[k][5], info[k][7], info[k][8], info[k][9], aggressive);
        const bool red = max_r[k] > max_b[k];
        const float up = delta / (nrmax / 2.f);
        up_r[k] = red ? up : 0.f;
        up_b[k] = red ? 0.f : up;
        dn_b[k] = red ? -(ch_M[k] - min_b) / nrmax : 0.f;
        dn_r[k] = red ? 0.f : -(ch_M[k] - min_r) / nrmax;
    }
    float chM = 0.f, top_r = 0.f, top_b = 0.f, low_r = 100000000000.f, low_b = 100000000000.f;
    float mean_up_r = 0.f, mean_up_b = 0.f, mean_dn_r = 0.f, mean_dn_b = 0.f;
    for (int k = 0; k < 9; ++k) {
        chM += ch_M[k]; mean_up_b += up_b[k]; mean_up_r += up_r[k]; mean_dn_r += dn_r[k]; mean_dn_b += dn_b[k];
        if (up_r[k] > top_r) top_r = up_r[k];
        if (up_b[k] > top_b) top_b = up_b[k];
        if (dn_r[k] < low_r) low_r = dn_r[k];
        if (dn_b[k] < low_b) low_b = dn_b[k];
    }
    chM /= 9; mean_up_b /= 9; mean_up_r /= 9; mean_dn_b /= 9; mean_dn_r /= 9;
    float maxr, maxb;
    if (top_r > top_b) {
        maxr = mean_up_r + (top_r - mean_up_r) * 0.66f;
        maxb = mean_dn_b + (low_b - mean_dn_b) * 0.66f;
    } else {
        maxb = mean_up_b + (top_b - mean_up_b) * 0.66f;
        maxr = mean_dn_r + (low_r - mean_dn_r) * 0.66f;
    }
    out3[0] = chM / nr; out3[1] = maxr; out3[2] = maxb;
}

} // namespace artgpu
