// art_amd/csrc/paramcurve.h -- DiagonalCurve::getVal for DCT_Parametric curves (reference: rtengine/diagonalcurves.cc:106-131,443-470,
// the elementary curves of rtengine/curves.h:92-156 and the double-precision sleef xlog / xexp they are written with, rtengine/sleef.h:
// 26-28,58-92,519-571).  curves::setLutVal (curves.h:224-231) sends every value above 65535 through the Curve object instead of the
// LUT; for a parametric curve that is this analytic form.  The same code runs on the host (the constructor's mc / mfc / msc / mhc) and
// on the device (per super-white pixel): IEEE double + - * / sqrt in the reference's order, contraction off.
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PC_HD __host__ __device__ inline
#else
#define PC_HD inline
#endif

namespace artgpu {

struct ParamCurve {
    double x[9];                 // as the constructor stores them (L113-125): x[1..3] clamped to [0.001, 0.99], x[4..7] = (p + 100) / 200, x[8] = p[8] / 100 or 1
    double mc, mfc, msc, mhc;    // L127-131
};

PC_HD int64_t pc_bits(double d) { int64_t i; memcpy(&i, &d, 8); return i; }
PC_HD double pc_from_bits(int64_t i) { double d; memcpy(&d, &i, 8); return d; }
PC_HD double pc_mla(double x, double y, double z) { return x * y + z; }
PC_HD double pc_xrint(double x) { return x < 0 ? (int)(x - 0.5) : (int)(x + 0.5); }
PC_HD double pc_ldexpk(double x, int q)
{
    int m = q >> 31;
    m = (((m + q) >> 9) - m) << 7;
    q = q - (m << 2);
    double u = pc_from_bits(((int64_t)(m + 0x3ff)) << 52);
    double u2 = u * u;
    u2 = u2 * u2;
    x = x * u2;
    u = pc_from_bits(((int64_t)(q + 0x3ff)) << 52);
    return x * u;
}
PC_HD int pc_ilogbp1(double d)
{
    const int m = d < 4.9090934652977266E-91;
    d = m ? 2.037035976334486E90 * d : d;
    int q = (int)((pc_bits(d) >> 52) & 0x7ff);
    q = m ? q - (300 + 0x03fe) : q - 0x03fe;
    return q;
}
PC_HD double pc_xlog(double d)
{
    const int e = pc_ilogbp1(d * 0.7071);
    const double m = pc_ldexpk(d, -e);
    double x = (m - 1) / (m + 1);
    const double x2 = x * x;
    double t = 0.148197055177935105296783;
    t = pc_mla(t, x2, 0.153108178020442575739679);
    t = pc_mla(t, x2, 0.181837339521549679055568);
    t = pc_mla(t, x2, 0.22222194152736701733275);
    t = pc_mla(t, x2, 0.285714288030134544449368);
    t = pc_mla(t, x2, 0.399999999989941956712869);
    t = pc_mla(t, x2, 0.666666666666685503450651);
    t = pc_mla(t, x2, 2);
    x = x * t + 0.693147180559945286226764 * e;
    if (d == INFINITY) x = INFINITY;
    if (d < 0) x = NAN;
    if (d == 0) x = -INFINITY;
    return x;
}
PC_HD double pc_xexp(double d)
{
    const int q = (int)pc_xrint(d * 1.442695040888963407359924681001892137426645954152985934135449406931);
    double s = pc_mla(q, -.69314718055966295651160180568695068359375, d);
    s = pc_mla(q, -.28235290563031577122588448175013436025525412068e-12, s);
    double u = 2.08860621107283687536341e-09;
    u = pc_mla(u, s, 2.51112930892876518610661e-08);
    u = pc_mla(u, s, 2.75573911234900471893338e-07);
    u = pc_mla(u, s, 2.75572362911928827629423e-06);
    u = pc_mla(u, s, 2.4801587159235472998791e-05);
    u = pc_mla(u, s, 0.000198412698960509205564975);
    u = pc_mla(u, s, 0.00138888888889774492207962);
    u = pc_mla(u, s, 0.00833333333331652721664984);
    u = pc_mla(u, s, 0.0416666666666665047591422);
    u = pc_mla(u, s, 0.166666666666666851703837);
    u = pc_mla(u, s, 0.5);
    u = s * s * u + s + 1;
    u = pc_ldexpk(u, q);
    if (d == -INFINITY) u = 0;
    return u;
}
// curves.h:92-156
PC_HD double pc_basel(double x, double m1, double m2)
{
    if (x == 0.0) return 0.0;
    const double k = sqrt((m1 - 1.0) * (m1 - m2) * 0.5) / (1.0 - m2);
    const double l = (m1 - m2) / (1.0 - m2) + k;
    const double lx = pc_xlog(x);
    return m2 * x + (1.0 - m2) * (2.0 - pc_xexp(k * lx)) * pc_xexp(l * lx);
}
PC_HD double pc_baseu(double x, double m1, double m2) { return 1.0 - pc_basel(1.0 - x, m1, m2); }
PC_HD double pc_cupper(double x, double m, double hr)
{
    if (hr > 1.0) return pc_baseu(x, m, 2.0 * (hr - 1.0) / m);
    const double x1 = (1.0 - hr) / m;
    const double x2 = x1 + hr;
    if (x >= x2) return 1.0;
    if (x < x1) return x * m;
    return 1.0 - hr + hr * pc_baseu((x - x1) / hr, m, 0);
}
PC_HD double pc_clower(double x, double m, double sr) { return 1.0 - pc_cupper(1.0 - x, m, sr); }
PC_HD double pc_p00(double x, double prot) { return pc_clower(x, 2.0, prot); }
PC_HD double pc_p11(double x, double prot) { return pc_cupper(x, 2.0, prot); }
PC_HD double pc_p01(double x, double prot) { return x <= 0.5 ? pc_clower(x * 2, 2.0, prot) * 0.5 : 0.5 + pc_cupper((x - 0.5) * 2, 2.0, prot) * 0.5; }
PC_HD double pc_p10(double x, double prot) { return x <= 0.5 ? pc_cupper(x * 2, 2.0, prot) * 0.5 : 0.5 + pc_clower((x - 0.5) * 2, 2.0, prot) * 0.5; }
PC_HD double pc_pfull(double x, double prot, double sh, double hl)
{
    return (1 - sh) * (1 - hl) * pc_p00(x, prot) + sh * hl * pc_p11(x, prot) + (1 - sh) * hl * pc_p01(x, prot) + sh * (1 - hl) * pc_p10(x, prot);
}
// the constructor's part (diagonalcurves.cc:106-131); p = the curve's parameter vector, p[0] = DCT_Parametric, np = 8 or 9
PC_HD void pc_init(ParamCurve &c, const double *p, int np)
{
    c.x[0] = p[0];
    for (int i = 1; i < 4; i++) { const double v = p[i] > 0.001 ? p[i] : 0.001; c.x[i] = v < 0.99 ? v : 0.99; }     // min(max(p[i], 0.001), 0.99)
    for (int i = 4; i < 8; i++) c.x[i] = (p[i] + 100.0) / 200.0;
    c.x[8] = np < 9 ? 1.0 : p[8] / 100.0;
    c.mc = -pc_xlog(2.0) / pc_xlog(c.x[2]);
    const double mbase = pc_pfull(0.5, c.x[8], c.x[6], c.x[5]);
    c.mfc = mbase <= 1e-14 ? 0.0 : pc_xexp(pc_xlog(mbase) / c.mc);
    c.msc = -pc_xlog(2.0) / pc_xlog(c.x[1] / c.x[2]);
    c.mhc = -pc_xlog(2.0) / pc_xlog((c.x[3] - c.x[2]) / (1 - c.x[2]));
}
// DiagonalCurve::getVal, case DCT_Parametric (diagonalcurves.cc:448-470)
PC_HD double pc_getval(const ParamCurve &c, double t)
{
    if (t <= 1e-14) return 0.0;
    const double tv = pc_xexp(c.mc * pc_xlog(t));
    const double base = pc_pfull(tv, c.x[8], c.x[6], c.x[5]);
    const double stretched = base <= 1e-14 ? 0.0 : pc_xexp(pc_xlog(base) / c.mc);
    if (t < c.x[2]) {
        const double stv = pc_xexp(c.msc * pc_xlog(stretched / c.mfc));
        const double sbase = pc_pfull(stv, c.x[8], c.x[7], 0.5);
        return c.mfc * (sbase <= 1e-14 ? 0.0 : pc_xexp(pc_xlog(sbase) / c.msc));
    }
    const double htv = pc_xexp(c.mhc * pc_xlog((stretched - c.mfc) / (1 - c.mfc)));
    const double hbase = pc_pfull(htv, c.x[8], 0.5, c.x[4]);
    return c.mfc + (1 - c.mfc) * (hbase <= 1e-14 ? 0.0 : pc_xexp(pc_xlog(hbase) / c.mhc));
}

} // namespace artgpu
