// art_amd/csrc/amaze.hip -- AMaZE demosaic for gfx950, v1 "arena" kernel.
//
// Replaces RawImageSource::amaze_demosaic_RT (reference: rtengine/amaze_demosaic_RT.cc:41-1595,
// x86-64 / __SSE2__ branches).  One workgroup processes one REFERENCE tile (160x160, origin
// (-16,-16), stride 128: the tile grid is part of the numerical result, SURVEY.md section 7
// rule 3) and walks the reference's phases with a workgroup barrier between them.
//
// Data layout: every workgroup owns a private work arena in HBM with the reference's plane
// order, gaps and aliasing (amaze_demosaic_RT.cc:124-174), so that aliased and out-of-range
// reads return what the reference's do.  The arena is zeroed per tile (a reference thread's
// first tile).  v1 keeps all planes in the arena (L2/MALL resident: 1.45 MB per workgroup);
// LDS only holds the tile-wide reduction words.
//
// In-place phases are parallelised without changing their results:
//   P3  hcd: lanes 2,3 of every 4-lane vector group only read original values, lanes 0,1 read
//       the previous group's updated lanes 2,3  -> two parallel sub-passes;
//       vcd: row rr reads the UPDATED row rr-2 -> one lane per column, sequential in rows;
//   P9  hvwt / P13 pmwt: row rr reads the updated row rr-1 -> row loop with a barrier per row.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "devmath.h"
#include "kernels.h"

namespace artgpu {

namespace {

constexpr int ts = 160, tsh = 80;
constexpr int v1 = ts, v2 = 2 * ts, v3 = 3 * ts;
constexpr int p1 = -ts + 1, p2 = -2 * ts + 2, p3 = -3 * ts + 3;
constexpr int m1 = ts + 1, m2 = 2 * ts + 2, m3 = 3 * ts + 3;
constexpr int F = ts * ts, Hh = ts * tsh, GAP = 32;

// arena offsets in floats (amaze_demosaic_RT.cc:124-174)
constexpr int O_rgbgreen = 0;
constexpr int O_delhvsqsum = O_rgbgreen + F + GAP;
constexpr int O_dirwts0 = O_delhvsqsum + F + GAP;
constexpr int O_dirwts1 = O_dirwts0 + F + GAP;
constexpr int O_vcd = O_dirwts1 + F + GAP;
constexpr int O_hcd = O_vcd + F + GAP;
constexpr int O_vcdalt = O_hcd + F + GAP;
constexpr int O_hcdalt = O_vcdalt + F + GAP;
constexpr int O_cddiffsq = O_hcdalt + F + GAP;
constexpr int O_hvwt = O_cddiffsq + F + 2 * GAP;
constexpr int O_dgintv = O_hvwt + Hh + GAP;
constexpr int O_dginth = O_dgintv + F + GAP;
constexpr int O_Dgrbsq1m = O_dginth + F + GAP;
constexpr int O_Dgrbsq1p = O_Dgrbsq1m + Hh + GAP;
constexpr int O_cfa = O_Dgrbsq1p + Hh + GAP;
constexpr int O_nyquist = O_cfa + F + GAP;           // bytes from here
constexpr int O_nyqutest = O_nyquist + Hh / 4 + GAP; // floats
constexpr int ARENA_FLOATS = ((O_nyqutest + Hh + 64) + 3) & ~3;
static_assert(ARENA_FLOATS == AMAZE_ARENA_FLOATS, "arena size mismatch with kernels.h");

constexpr float eps = 1e-5f, epssq = 1e-10f, arthresh = 0.75f;

// Items (row rr, index it) of a phase, strided over the workgroup.  (rr, it) advance incrementally: one integer division per phase
// and thread instead of one per item (the divisor is a run-time value; the division was a fifth of the kernel's instructions).
#define FOR_ITEMS(R0, R1, N)                                                                                                      \
    for (int _n = (N), _tot = ((R1) > (R0) ? ((R1) - (R0)) : 0) * _n, _t = tid, _d = _n > 0 ? _n : 1, _q = NTT / _d, _r = NTT - _q * _d, \
             rr = (R0) + tid / _d, it = tid - (rr - (R0)) * _d;                                                                  \
         _t < _tot; _t += NTT, rr += _q, it += _r, rr += (it >= _d), it -= (it >= _d) ? _d : 0)                                   \
        for (int _once = 1; _once; _once = 0)

// highlight bounding of a colour difference (amaze_demosaic_RT.cc:555-581)
__device__ __forceinline__ float bound_cd(float cdv, float sgn, float c, float nA, float nB, float clip_pt)
{
    const float nsgn = -sgn, sgn3 = sgn + sgn + sgn;
    const float Gint = sgn * cdv + c;
    const float temp2 = sgn3 * cdv;
    const float wt = 1.f + temp2 / (eps + Gint + c);
    const bool mask = (nsgn * cdv) > 0.f;
    const float old = cdv;
    const float temp = nsgn * (c - median3(Gint, nA, nB));
    cdv = (temp2 < -(c + Gint)) ? temp : intp(wt, cdv, temp);
    cdv = mask ? cdv : old;
    cdv = (Gint > clip_pt) ? temp : cdv;
    return cdv;
}

__device__ __forceinline__ float var3(float a, float b, float c)
{
    return sqr(a - b) + sqr(a - c) + sqr(b - c);
}

// diagonal R/B estimate in one direction pair (amaze_demosaic_RT.cc:1064-1084)
__device__ __forceinline__ float rb_ratio(float cfav, float t1, float t2)
{
    float r = (t1 + t1) / (eps + cfav + t2);
    return fabsf(1.f - r) < arthresh ? cfav * r : t1 + 0.5f * (cfav - t2);
}
__device__ __forceinline__ float rb_bound(float rbv, float cfav, float nA, float nB, float clip_pt)
{
    float t1 = median3(rbv, nA, nB);
    float wt = ((cfav - rbv) + (cfav - rbv)) / (eps + rbv + cfav);
    float t2 = intp(wt, rbv, t1);
    t2 = (rbv + rbv < cfav) ? t1 : t2;
    t2 = (rbv < cfav) ? t2 : rbv;
    return (t2 > clip_pt) ? median3(t2, nA, nB) : t2;
}
// G from R+B in one axis (amaze_demosaic_RT.cc:1253-1289)
__device__ __forceinline__ float g_dir(float rb, float cn, float rbn)
{
    float cr = (cn + cn) / (eps + rb + rbn);
    float g = rb * cr;
    float g2 = cn + 0.5f * (rb - rbn);
    return fabsf(1.f - cr) < arthresh ? g : g2;
}
__device__ __forceinline__ float g_bound(float Gint, float rb, float nA, float nB, float clip_pt)
{
    float G1 = median3(Gint, nA, nB);
    float wt = ((rb - Gint) + (rb - Gint)) / (eps + Gint + rb);
    float G2 = intp(wt, Gint, G1);
    G1 = ((Gint + Gint) < rb) ? G1 : G2;
    Gint = (Gint < rb) ? G1 : Gint;
    return (Gint > clip_pt) ? median3(Gint, nA, nB) : Gint;
}

} // namespace


#ifndef AMAZE_MIN_WAVES
#define AMAZE_MIN_WAVES 6   // fused kernel: 512 threads x 3 workgroups per CU (51 KB LDS plane each) = 6 waves per SIMD: <= 80 VGPRs (some spill); measured best of {256,384,512,640,768,1024} x {3..8}
#endif
// LDS plane of the row-recurrence / site-list phases: allocated only in the kernels that contain one of them
template <bool NEED>
__device__ __forceinline__ float *lds_plane()
{
    if constexpr (NEED) {
        __shared__ float buf[ts * tsh];
        return buf;
    } else {
        return nullptr;
    }
}
constexpr int AMAZE_NPHASES = 20;

// The tile algorithm as 20 phases.  amaze_kernel<FIRST, LAST, G, MINW> runs phases FIRST..LAST of one tile with G workgroups
// per tile (G > 1 only for phases that are plain parallel loops).  <0, 19, 1, 6> is the fused one-workgroup-per-tile kernel
// (barriers between phases, the default); <k, k, G, 1> is phase k alone over all tiles (ARTGPU_AMAZE_SPLIT=1: one launch per
// phase, every tile keeps its arena in HBM between launches) -- same results, used to profile the phases individually.
template <int FIRST, int LAST, int G, int MINW>
__global__ void __launch_bounds__(AMAZE_THREADS, MINW)
amaze_kernel(AmazeArgs a)
{
#define RUN(k) ((k) >= FIRST && (k) <= LAST)
#define SYNC_AFTER(k) do { if constexpr (RUN(k) && RUN((k) + 1)) __syncthreads(); } while (0)
    constexpr int NTT = AMAZE_THREADS * G;
    const int tid = threadIdx.x + (G > 1 ? (int)(blockIdx.x % G) * AMAZE_THREADS : 0);
    float *const A = a.arena + (size_t)(blockIdx.x / G) * AMAZE_ARENA_FLOATS;
    int *const bbox = a.bbox + (size_t)(blockIdx.x / G) * 4;   // nyquist bounding box of the tile (min row, max row, min col, max col)
    float *const s_plane = lds_plane<RUN(10) || RUN(11) || RUN(15)>();
    float *const rgbgreen = A + O_rgbgreen, *const delhvsqsum = A + O_delhvsqsum;
    float *const dirwts0 = A + O_dirwts0, *const dirwts1 = A + O_dirwts1;
    float *const vcd = A + O_vcd, *const hcd = A + O_hcd, *const vcdalt = A + O_vcdalt, *const hcdalt = A + O_hcdalt;
    float *const cddiffsq = A + O_cddiffsq, *const hvwt = A + O_hvwt;
    float *const Dgrb0 = vcdalt, *const Dgrb1 = vcdalt + Hh;
    float *const delp = cddiffsq, *const delm = delp + Hh + GAP, *const rbint = delm;
    float *const dgintv = A + O_dgintv, *const dginth = A + O_dginth, *const Dgrb2 = dgintv;
    float *const Dgrbsq1m = A + O_Dgrbsq1m, *const Dgrbsq1p = A + O_Dgrbsq1p;
    float *const cfa = A + O_cfa;
    float *const pmwt = delhvsqsum, *const rbm = vcd, *const rbp = rbm + Hh + GAP;
    unsigned char *const nyquist = reinterpret_cast<unsigned char *>(A + O_nyquist);
    unsigned char *const nyquist2 = reinterpret_cast<unsigned char *>(cddiffsq);
    float *const nyqutest = A + O_nyqutest;

    const unsigned filters = a.filters;
    const int width = a.W, height = a.H;
    const float clip_pt = a.clip_pt, clip_pt8 = a.clip_pt8;
    const float *const raw = a.raw;
    const size_t rs = a.raw_stride;

    __shared__ int s_red[5]; // min row, max row, min col, max col of nyquist flags; [4] = P8 site count
    int &s_count = s_red[4];

    int ex, ey;
    if (fc(filters, 0, 0) == 1) {
        if (fc(filters, 0, 1) == 0) { ey = 0; ex = 1; } else { ey = 1; ex = 0; }
    } else {
        if (fc(filters, 0, 0) == 0) { ey = 0; ex = 0; } else { ey = 1; ex = 1; }
    }

    const int nlist = a.tile_count ? *a.tile_count : a.ntiles;
    const int qtaken = a.queue_hdr ? a.queue_hdr[1] : 0, qleft = a.queue_hdr ? max(a.queue_hdr[0] - qtaken, 0) : 0;
    const int nlisted = nlist + qleft;
    if (a.queue_hdr && blockIdx.x == 0 && threadIdx.x == 0 && a.queue_counters) { a.queue_counters[4] = nlist; a.queue_counters[5] = qtaken; a.queue_counters[6] = a.queue_hdr[0]; }
    for (int kt = blockIdx.x / G; kt < nlisted; kt += gridDim.x / G) {
        const int tile = kt >= nlist ? (int)(a.queue_words[qtaken + kt - nlist] & 0xffffffu) : (a.tile_list ? a.tile_list[kt] : kt);
        const int ty = tile / a.ntx, tx = tile - ty * a.ntx;
        const int top = -16 + ty * (ts - 32), left = -16 + tx * (ts - 32);
        const int bottom = min(top + ts, height + 16), right = min(left + ts, width + 16);
        const int rr1 = bottom - top, cc1 = right - left;
        const int rrmin = top < 0 ? 16 : 0, ccmin = left < 0 ? 16 : 0;
        const int rrmax = bottom > height ? height - top : rr1;
        const int ccmax = right > width ? width - left : cc1;

        // ======== phase 0: clear what has to be cleared, tile initialisation ========
        if constexpr (RUN(0)) {
        // ---- zero the arena (fresh-calloc semantics) ----
        {
            constexpr int NREG = 17;
            const int reg_off[NREG + 1] = {O_rgbgreen, O_delhvsqsum, O_dirwts0, O_dirwts1, O_vcd, O_hcd, O_vcdalt, O_hcdalt, O_cddiffsq, O_hvwt, O_dgintv,
                                           O_dginth, O_Dgrbsq1m, O_Dgrbsq1p, O_cfa, O_nyquist, O_nyqutest, ARENA_FLOATS};
            const unsigned zmask = (rr1 == ts && cc1 == ts) ? a.zero_mask : 0xffffffffu;   // partial tiles: everything
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < NREG; ++r)
                if ((zmask >> r) & 1u) {
                    float4 *A4 = reinterpret_cast<float4 *>(A + reg_off[r]);
                    const int kf = a.zero_frame;
                    if (kf > 0 && zmask != 0xffffffffu && r >= 4 && r <= 8) {
                        // full-size float planes: only a frame of kf rows / columns (and the gap behind the plane) is read before it is written
                        const int nrow4 = kf * ts / 4, tail0 = (ts - kf) * ts / 4, tail1 = (reg_off[r + 1] - reg_off[r]) / 4;
                        for (int i = tid; i < nrow4; i += NTT) A4[i] = z;
                        for (int i = tail0 + tid; i < tail1; i += NTT) A4[i] = z;
                        float *P = A + reg_off[r];
                        for (int i = tid; i < (ts - 2 * kf) * 2 * kf; i += NTT) {
                            const int rr = kf + i / (2 * kf), c = i % (2 * kf);
                            P[rr * ts + (c < kf ? c : ts - 2 * kf + c)] = 0.f;
                        }
                    } else
                        for (int i = tid; i < (reg_off[r + 1] - reg_off[r]) / 4; i += NTT) A4[i] = z;
                }
            if (tid == 0) { bbox[0] = 1 << 30; bbox[1] = 0; bbox[2] = ts + 1; bbox[3] = 0; }
        }
        __syncthreads();

        // ---- tile initialisation (L205-334), in the reference's write order ----
#define SETCFA(i, v) do { float t_ = (v) / 65535.f; cfa[i] = t_; rgbgreen[i] = t_; } while (0)
#define RAW(r, c) raw[(size_t)(r) * rs + (c)]
        if (rrmin > 0) {
            const int n = ccmax - ccmin;
            for (int t = tid; t < 16 * n; t += NTT) { int rr = t / n, cc = ccmin + t - rr * n; SETCFA(rr * ts + cc, RAW(32 - rr + top, cc + left)); }
        }
        {
            const int n = ccmax - ccmin, nr = rrmax - rrmin;
            FOR_ITEMS(rrmin, rrmin + nr, n > 0 ? n : 0) { const int cc = ccmin + it; SETCFA(rr * ts + cc, RAW(rr + top, cc + left)); }
        }
        __syncthreads();
        if (rrmax < rr1) {
            const int n = ccmax - ccmin;
            for (int t = tid; t < 16 * n; t += NTT) { int rr = t / n, cc = ccmin + t - rr * n; SETCFA((rrmax + rr) * ts + cc, RAW(height - rr - 2, left + cc)); }
        }
        __syncthreads();
        if (ccmin > 0) {
            const int nr = rrmax - rrmin;
            for (int t = tid; t < nr * 16; t += NTT) { int rr = rrmin + (t >> 4), cc = t & 15; SETCFA(rr * ts + cc, RAW(rr + top, 32 - cc + left)); }
        }
        __syncthreads();
        if (ccmax < cc1) {
            const int nr = rrmax - rrmin;
            for (int t = tid; t < nr * 16; t += NTT) { int rr = rrmin + (t >> 4), cc = t & 15; SETCFA(rr * ts + ccmax + cc, RAW(top + rr, width - cc - 2)); }
        }
        __syncthreads();
        if (tid < 256) {
            const int rr = tid >> 4, cc = tid & 15;
            if (rrmin > 0 && ccmin > 0) SETCFA(rr * ts + cc, RAW(32 - rr, 32 - cc));
        }
        __syncthreads();
        if (tid < 256) {
            const int rr = tid >> 4, cc = tid & 15;
            if (rrmax < rr1 && ccmax < cc1) SETCFA((rrmax + rr) * ts + ccmax + cc, RAW(height - rr - 2, width - cc - 2));
        }
        __syncthreads();
        if (tid < 256) {
            const int rr = tid >> 4, cc = tid & 15;
            if (rrmin > 0 && ccmax < cc1) SETCFA(rr * ts + ccmax + cc, RAW(32 - rr, width - cc - 2));
        }
        __syncthreads();
        if (tid < 256) {
            const int rr = tid >> 4, cc = tid & 15;
            if (rrmax < rr1 && ccmin > 0) SETCFA((rrmax + rr) * ts + cc, RAW(height - rr - 2, 32 - cc));
        }
#undef SETCFA
#undef RAW
        }
        SYNC_AFTER(0);

        // ---- P1: gradients (L342-351); 4-lane groups over [0, cc1) ----
        if constexpr (RUN(1))
        FOR_ITEMS(2, rr1 - 2, 4 * ngroups(0, cc1, 4)) {
            const int i = rr * ts + it;
            const float c0 = cfa[i];
            const float delh = fabsf(cfa[i + 1] - cfa[i - 1]);
            const float delv = fabsf(cfa[i + v1] - cfa[i - v1]);
            dirwts1[i] = eps + fabsf(cfa[i + 2] - c0) + fabsf(c0 - cfa[i - 2]) + delh;
            dirwts0[i] = eps + fabsf(cfa[i + v2] - c0) + fabsf(c0 - cfa[i - v2]) + delv;
            delhvsqsum[i] = sqr(delh) + sqr(delv);
        }
        SYNC_AFTER(1);

        // ---- P2: vertical/horizontal colour differences (L380-434) ----
        if constexpr (RUN(2))
        FOR_ITEMS(4, rr1 - 4, 4 * ngroups(4, cc1 - 7, 4)) {
            const int cc = 4 + it, i = rr * ts + cc;
            const float sgn = (fc(filters, rr, cc) & 1) ? -1.f : 1.f;
            const float cfav = cfa[i];
            const float cu1 = cfa[i - v1], cu2 = cfa[i - v2], cd1 = cfa[i + v1], cd2 = cfa[i + v2];
            const float cl1 = cfa[i - 1], cl2 = cfa[i - 2], cr1 = cfa[i + 1], cr2 = cfa[i + 2];
            const float d0c = dirwts0[i], d1c = dirwts1[i];
            const float d0u2 = dirwts0[i - v2], d0d2 = dirwts0[i + v2], d1l2 = dirwts1[i - 2], d1r2 = dirwts1[i + 2];
            const float cru = cu1 * (d0u2 + d0c) / (d0u2 * (eps + cfav) + d0c * (eps + cu2));
            const float crd = cd1 * (d0d2 + d0c) / (d0d2 * (eps + cfav) + d0c * (eps + cd2));
            const float crl = cl1 * (d1l2 + d1c) / (d1l2 * (eps + cfav) + d1c * (eps + cl2));
            const float crr = cr1 * (d1r2 + d1c) / (d1r2 * (eps + cfav) + d1c * (eps + cr2));
            const float guha = cu1 + 0.5f * (cfav - cu2);
            const float gdha = cd1 + 0.5f * (cfav - cd2);
            const float glha = cl1 + 0.5f * (cfav - cl2);
            const float grha = cr1 + 0.5f * (cfav - cr2);
            float guar = fabsf(1.f - cru) < arthresh ? cfav * cru : guha;
            float gdar = fabsf(1.f - crd) < arthresh ? cfav * crd : gdha;
            float glar = fabsf(1.f - crl) < arthresh ? cfav * crl : glha;
            float grar = fabsf(1.f - crr) < arthresh ? cfav * crr : grha;
            const float d1l = dirwts1[i - 1], d1r = dirwts1[i + 1], d0u = dirwts0[i - v1], d0d = dirwts0[i + v1];
            const float hwt = d1l / (d1l + d1r);
            const float vwt = d0u / (d0d + d0u);
            const float Ginthha = intp(hwt, grha, glha);
            const float Gintvha = intp(vwt, gdha, guha);
            const float hcdaltv = sgn * (Ginthha - cfav);
            const float vcdaltv = sgn * (Gintvha - cfav);
            hcdalt[i] = hcdaltv;
            vcdalt[i] = vcdaltv;
            const bool clip = (cfav > clip_pt8) || (Gintvha > clip_pt8) || (Ginthha > clip_pt8);
            if (clip) { guar = guha; gdar = gdha; glar = glha; grar = grha; }
            vcd[i] = clip ? vcdaltv : sgn * (intp(vwt, gdar, guar) - cfav);
            hcd[i] = clip ? hcdaltv : sgn * (intp(hwt, grar, glar) - cfav);
            dgintv[i] = sse_min(sqr(guha - gdha), sqr(guar - gdar));
            dginth[i] = sse_min(sqr(glha - grha), sqr(glar - grar));
        }
        SYNC_AFTER(2);

        // ---- P3: variance choice + highlight bounding, in place in the reference (L540-583) ----
        const int ng3 = ngroups(4, cc1 - 4, 4);
        float *const Thi = Dgrbsq1m; // new hcd of lanes 2,3 (plane is free until P11)
        float *const Tlo = Dgrbsq1p; // new hcd of lanes 0,1
        // 3a: lanes 2,3 read only original hcd
        if constexpr (RUN(3))
        FOR_ITEMS(4, rr1 - 4, 2 * ng3) {
            const int g = it >> 1, k = 2 + (it & 1), cc = 4 + 4 * g + k, i = rr * ts + cc;
            const float sgn = (fc(filters, rr, cc) & 1) ? -1.f : 1.f;
            float hcdv = hcd[i];
            const float hv = var3(hcd[i - 2], hcdv, hcd[i + 2]);
            const float ha = hcdalt[i];
            const float hav = var3(hcdalt[i - 2], ha, hcdalt[i + 2]);
            hcdv = hav < hv ? ha : hcdv;
            Thi[rr * tsh + it] = bound_cd(hcdv, sgn, cfa[i], cfa[i - 1], cfa[i + 1], clip_pt);
        }
        SYNC_AFTER(3);
        // 3b: lanes 0,1 read the previous group's updated lanes 2,3 at i-2
        if constexpr (RUN(4)) {
        FOR_ITEMS(4, rr1 - 4, 2 * ng3) {
            const int g = it >> 1, k = it & 1, cc = 4 + 4 * g + k, i = rr * ts + cc;
            const float sgn = (fc(filters, rr, cc) & 1) ? -1.f : 1.f;
            const float hm2 = g > 0 ? Thi[rr * tsh + 2 * (g - 1) + k] : hcd[i - 2];
            float hcdv = hcd[i];
            const float hv = var3(hm2, hcdv, hcd[i + 2]);
            const float ha = hcdalt[i];
            const float hav = var3(hcdalt[i - 2], ha, hcdalt[i + 2]);
            hcdv = hav < hv ? ha : hcdv;
            Tlo[rr * tsh + it] = bound_cd(hcdv, sgn, cfa[i], cfa[i - 1], cfa[i + 1], clip_pt);
        }
        // 3c: vcd, one lane per column, rows in order (row rr reads the updated row rr-2)
        for (int t = tid; t < 4 * ng3; t += NTT) {
            const int cc = 4 + t;
            float n2 = vcd[2 * ts + cc], n1 = vcd[3 * ts + cc];
            float o0 = vcd[4 * ts + cc], o1 = vcd[5 * ts + cc];
            float am2 = vcdalt[2 * ts + cc], am1 = vcdalt[3 * ts + cc], a0 = vcdalt[4 * ts + cc], a1 = vcdalt[5 * ts + cc];
            float cm1 = cfa[3 * ts + cc], c0 = cfa[4 * ts + cc];
            for (int rr = 4; rr < rr1 - 4; ++rr) {
                const int i = rr * ts + cc;
                const float sgn = (fc(filters, rr, cc) & 1) ? -1.f : 1.f;
                const float o2 = vcd[i + v2], a2 = vcdalt[i + v2], cp1 = cfa[i + v1];
                float vcdv = o0;
                const float vv = var3(n2, vcdv, o2);
                const float vav = var3(am2, a0, a2);
                vcdv = vav < vv ? a0 : vcdv;
                const float nv = bound_cd(vcdv, sgn, c0, cm1, cp1, clip_pt);
                vcd[i] = nv;
                n2 = n1; n1 = nv; o0 = o1; o1 = o2;
                am2 = am1; am1 = a0; a0 = a1; a1 = a2;
                cm1 = c0; c0 = cp1;
            }
        }
        }
        SYNC_AFTER(4);
        // 3d: commit hcd, cddiffsq
        if constexpr (RUN(5))
        FOR_ITEMS(4, rr1 - 4, 4 * ng3) {
            const int g = it >> 2, k = it & 3, cc = 4 + it, i = rr * ts + cc;
            const float h = (k < 2) ? Tlo[rr * tsh + 2 * g + k] : Thi[rr * tsh + 2 * g + (k - 2)];
            hcd[i] = h;
            cddiffsq[i] = sqr(vcd[i] - h);
        }
        SYNC_AFTER(5);

        // ---- P4: h/v weight at R/B sites (L680-728) ----
        if constexpr (RUN(6)) {
        FOR_ITEMS(6, rr1 - 6, 4 * ngroups(6, cc1 - 6, 8)) {
            const int par = fc(filters, rr, 2) & 1;
            if (it < 4 * ngroups(6 + par, cc1 - 6, 8)) {
                const int i = rr * ts + 6 + par + 2 * it;
                float t = vcd[i];
                const float vu1 = vcd[i - v1], vu2 = vcd[i - v2], vu3 = vcd[i - v3];
                const float vd1 = vcd[i + v1], vd2 = vcd[i + v2], vd3 = vcd[i + v3];
                const float uave = t + vu1 + vu2 + vu3;
                const float dave = t + vd1 + vd2 + vd3;
                float Dvu = sqr(t - uave) + sqr(vu1 - uave) + sqr(vu2 - uave) + sqr(vu3 - uave);
                float Dvd = sqr(t - dave) + sqr(vd1 - dave) + sqr(vd2 - dave) + sqr(vd3 - dave);
                const float d1l = dirwts1[i - 1], d1r = dirwts1[i + 1], d0u = dirwts0[i - v1], d0d = dirwts0[i + v1];
                const float hwt = d1l / (d1l + d1r);
                const float vwt = d0u / (d0u + d0d);
                t = hcd[i];
                const float hl1 = hcd[i - 1], hl2 = hcd[i - 2], hl3 = hcd[i - 3];
                const float hr1 = hcd[i + 1], hr2 = hcd[i + 2], hr3 = hcd[i + 3];
                const float lave = t + (hl3 + hl2) + hl1;
                const float rave = t + (hr1 + hr2) + hr3;
                float Dhl = sqr(t - lave) + sqr(hl1 - lave) + sqr(hl2 - lave) + sqr(hl3 - lave);
                float Dhr = sqr(t - rave) + sqr(hr1 - rave) + sqr(hr2 - rave) + sqr(hr3 - rave);
                const float vcdvar = epssq + intp(vwt, Dvd, Dvu);
                const float hcdvar = epssq + intp(hwt, Dhr, Dhl);
                Dvu = dgintv[i - v1] + dgintv[i - v2];
                Dvd = dgintv[i + v1] + dgintv[i + v2];
                Dhl = dginth[i - 2] + dginth[i - 1];
                Dhr = dginth[i + 1] + dginth[i + 2];
                const float vcdvar1 = epssq + dgintv[i] + intp(vwt, Dvd, Dvu);
                const float hcdvar1 = epssq + dginth[i] + intp(hwt, Dhr, Dhl);
                const float varwt = hcdvar / (vcdvar + hcdvar);
                const float diffwt = hcdvar1 / (vcdvar1 + hcdvar1);
                const bool dec = ((0.5f - varwt) * (0.5f - diffwt) > 0.f) && (fabsf(0.5f - diffwt) < fabsf(0.5f - varwt));
                hvwt[i >> 1] = dec ? varwt : diffwt;
            }
        }

        // ---- P5: nyquist test value (L746-803); vector groups then scalar tail ----
        {
            const float go0 = 0.14659727707323927f, go1 = 0.103592713382435f, go2 = 0.0732036125103057f, go3 = 0.0365543548389495f;
            const float nyqthresh = 0.5f;
            const float gg0 = nyqthresh * 0.07384411893421103f, gg1 = nyqthresh * 0.06207511968171489f, gg2 = nyqthresh * 0.0521818194747806f;
            const float gg3 = nyqthresh * 0.03687419286733595f, gg4 = nyqthresh * 0.03099732204057846f, gg5 = nyqthresh * 0.018413194161458882f;
            FOR_ITEMS(6, rr1 - 6, 4 * ngroups(6, cc1 - 7, 8) + 4) {
                const int par = fc(filters, rr, 2) & 1;
                const int nvec = 4 * ngroups(6 + par, cc1 - 7, 8);
                const int cc = 6 + par + 2 * it;
                const bool vec = it < nvec;
                if (vec || cc < cc1 - 6) {
                    const int i = rr * ts + cc;
                    const float *c = cddiffsq, *d = delhvsqsum;
                    const float gA = go0 * c[i] +
                                     go1 * (c[i - m1] + c[i + p1] + c[i - p1] + c[i + m1]) +
                                     go2 * (c[i - v2] + c[i - 2] + c[i + 2] + c[i + v2]) +
                                     go3 * (c[i - m2] + c[i + p2] + c[i - p2] + c[i + m2]);
                    const float s1 = vec ? (d[i - v1] + d[i - 1] + d[i + 1] + d[i + v1])
                                         : (d[i - v1] + d[i + 1] + d[i - 1] + d[i + v1]);
                    const float gB = gg0 * d[i] + gg1 * s1 +
                                     gg2 * (d[i - m1] + d[i + p1] + d[i - p1] + d[i + m1]) +
                                     gg3 * (d[i - v2] + d[i - 2] + d[i + 2] + d[i + v2]) +
                                     gg4 * (d[i - v2 - 1] + d[i - v2 + 1] + d[i - ts - 2] + d[i - ts + 2] +
                                            d[i + ts - 2] + d[i + ts + 2] + d[i + v2 - 1] + d[i + v2 + 1]) +
                                     gg5 * (d[i - m2] + d[i + p2] + d[i - p2] + d[i + m2]);
                    nyqutest[i >> 1] = gA - gB;
                }
            }
        }
        }
        SYNC_AFTER(6);

        // ---- P6: nyquist flags + bounding box (L806-825): per-workgroup box in LDS, merged into the tile's box in HBM ----
        if constexpr (RUN(7)) {
        if (threadIdx.x == 0) { s_red[0] = 1 << 30; s_red[1] = 0; s_red[2] = ts + 1; s_red[3] = 0; }
        __syncthreads();
        FOR_ITEMS(6, rr1 - 6, ngroups(6, cc1 - 6, 2)) {
            const int cc = 6 + (fc(filters, rr, 2) & 1) + 2 * it;
            if (cc < cc1 - 6) {
                const int i = rr * ts + cc;
                if (nyqutest[i >> 1] > 0.f) {
                    nyquist[i >> 1] = 1;
                    atomicMin(&s_red[0], rr);
                    atomicMax(&s_red[1], rr);
                    atomicMin(&s_red[2], cc);
                    atomicMax(&s_red[3], cc);
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicMin(&bbox[0], s_red[0]); atomicMax(&bbox[1], s_red[1]);
            atomicMin(&bbox[2], s_red[2]); atomicMax(&bbox[3], s_red[3]);
        }
        }
        SYNC_AFTER(7);
        // the tile's box: in a kernel that starts after phase 7 it comes from HBM (all workgroups of phase 7 have finished);
        // in the fused kernel the single workgroup's LDS copy already is the tile's box
        int nystartrow = 0, nyendrow = 0, nystartcol = 0, nyendcol = 0;
        bool doNyquist = false;
        if constexpr (LAST >= 8) {
            if constexpr (FIRST >= 8) {
                if (threadIdx.x < 4) s_red[threadIdx.x] = bbox[threadIdx.x];
                __syncthreads();
            }
            nystartrow = s_red[0] == (1 << 30) ? 0 : s_red[0];
            nyendrow = s_red[1]; nystartcol = s_red[2]; nyendcol = s_red[3];
            doNyquist = nystartrow != nyendrow && nystartcol != nyendcol;
            if (doNyquist) {
                nyendrow++;
                nyendcol++;
                nystartcol -= (nystartcol & 1);
                nystartrow = max(8, nystartrow);
                nyendrow = min(rr1 - 8, nyendrow);
                nystartcol = max(8, nystartcol);
                nyendcol = min(cc1 - 8, nyendcol);
            }
        }
        // ---- phase 8: memset(&nyquist2[4*tsh], 0, (ts-8)*tsh) (L879)
        if constexpr (RUN(8)) {
            if (doNyquist) {
                unsigned *w = reinterpret_cast<unsigned *>(nyquist2 + 4 * tsh);
                for (int t = tid; t < (ts - 8) * tsh / 4; t += NTT) w[t] = 0u;
            }
        }
        SYNC_AFTER(8);
        if constexpr (RUN(9)) {
            if (doNyquist)
            // ---- P7: majority vote with byte offsets independent of the row parity (L888-901) ----
            // four flags per item through aligned 32-bit accesses (byte-wise global loads/stores made this the slowest
            // phase of the tile); per-byte logic unchanged
            FOR_ITEMS(nystartrow, nyendrow, 4 * ngroups(0, cc1, 32)) {
                const int b = (rr * ts >> 1) + 4 * it;
                const unsigned *n4 = reinterpret_cast<const unsigned *>(nyquist + b);   // b and ts/2 are multiples of 4
                const unsigned up = n4[-(ts / 4)], dn = n4[ts / 4];
                const unsigned long long U = (unsigned long long)(n4[-21] >> 24) | ((unsigned long long)n4[-20] << 8);      // bytes b-81 .. b-77
                const unsigned long long M = (unsigned long long)(n4[-1] >> 24) | ((unsigned long long)n4[0] << 8) | ((unsigned long long)(n4[1] & 0xffu) << 40);   // b-1 .. b+4
                const unsigned long long D = (unsigned long long)(n4[19] >> 24) | ((unsigned long long)n4[20] << 8);        // bytes b+79 .. b+83
                unsigned outw = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int tsum = (int)((up >> (8 * k)) & 0xff) + (int)((U >> (8 * k)) & 0xff) + (int)((U >> (8 * k + 8)) & 0xff) + (int)((M >> (8 * k)) & 0xff) +
                                     (int)((M >> (8 * k + 16)) & 0xff) + (int)((D >> (8 * k)) & 0xff) + (int)((D >> (8 * k + 8)) & 0xff) + (int)((dn >> (8 * k)) & 0xff);
                    unsigned val = (unsigned)((M >> (8 * k + 8)) & 0xff);
                    if (tsum > 4) val = 1;
                    if (tsum < 4) val = 0;
                    outw |= val << (8 * k);
                }
                *reinterpret_cast<unsigned *>(nyquist2 + b) = outw;
            }
        }
        SYNC_AFTER(9);
        if constexpr (RUN(10)) {
          if (doNyquist) {      // uniform over the workgroup
            // ---- P8: area interpolation (L914-951).  Flagged sites are sparse: compact them into an LDS list first so
            // that the 7x7 gather loop runs with full waves instead of once per wave that contains a flag ----
            int *const s_list = reinterpret_cast<int *>(s_plane);
            if (tid == 0) s_count = 0;
            __syncthreads();
            FOR_ITEMS(nystartrow, nyendrow, ngroups(nystartcol, nyendcol, 2)) {
                const int cc = nystartcol + (fc(filters, rr, 2) & 1) + 2 * it;
                const int i = rr * ts + cc;
                if (cc < nyendcol && nyquist2[i >> 1]) s_list[atomicAdd(&s_count, 1)] = i;   // at most (ts-16)*(ts-16)/2 entries
            }
            __syncthreads();
            for (int q = tid, nq = s_count; q < nq; q += NTT) {
                const int i = s_list[q];
                float sumcfa = 0.f, sumh = 0.f, sumv = 0.f, sumsqh = 0.f, sumsqv = 0.f, areawt = 0.f;
                for (int ai = -6; ai < 7; ai += 2) {
                    int i1 = i + ai * ts - 6;
                    for (int bj = -6; bj < 7; bj += 2, i1 += 2) {
                        if (nyquist2[i1 >> 1]) {
                            const float ct = cfa[i1];
                            sumcfa += ct;
                            sumh += (cfa[i1 - 1] + cfa[i1 + 1]);
                            sumv += (cfa[i1 - v1] + cfa[i1 + v1]);
                            sumsqh += sqr(ct - cfa[i1 - 1]) + sqr(ct - cfa[i1 + 1]);
                            sumsqv += sqr(ct - cfa[i1 - v1]) + sqr(ct - cfa[i1 + v1]);
                            areawt += 1.f;
                        }
                    }
                }
                sumh = sumcfa - xdiv2f(sumh);
                sumv = sumcfa - xdiv2f(sumv);
                areawt = xdiv2f(areawt);
                const float hcdvar = epssq + fabsf(areawt * sumsqh - sumh * sumh);
                const float vcdvar = epssq + fabsf(areawt * sumsqv - sumv * sumv);
                hvwt[i >> 1] = hcdvar / (vcdvar + hcdvar);
            }
          }
        }
        SYNC_AFTER(10);

        // ---- P9: hvwt refined in place, row by row; G at R/B sites (L957-974) ----
        // The row recurrence only involves hvwt (row rr reads the updated row rr-1 and the old row rr+1): the half-resolution
        // plane is staged in LDS, ONE wave walks the rows with wave-level ordering only (no workgroup barrier per row), and
        // everything that merely consumes the refined weight runs afterwards as a parallel pass.
        if constexpr (RUN(11)) {
        for (int t = tid; t < ts * tsh; t += NTT) s_plane[t] = hvwt[t];
        __syncthreads();
        if (tid < 64) {
            for (int rr = 8; rr < rr1 - 8; ++rr) {
                const int par = fc(filters, rr, 2) & 1;
                for (int j = tid; j < 72; j += 64) {
                    const int cc = 8 + par + 2 * j;
                    if (cc < cc1 - 8) {
                        const int i = rr * ts + cc;
                        const float hvwtalt = xdivf(s_plane[(i - m1) >> 1] + s_plane[(i + p1) >> 1] + s_plane[(i - p1) >> 1] + s_plane[(i + m1) >> 1], 2);
                        const float h0 = s_plane[i >> 1];
                        s_plane[i >> 1] = fabsf(0.5f - h0) < fabsf(0.5f - hvwtalt) ? hvwtalt : h0;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        FOR_ITEMS(8, rr1 - 8, 72) {
            const int cc = 8 + (fc(filters, rr, 2) & 1) + 2 * it;
            if (cc < cc1 - 8) {
                const int i = rr * ts + cc;
                const float h = s_plane[i >> 1];
                hvwt[i >> 1] = h;
                const float dg = intp(h, vcd[i], hcd[i]);
                Dgrb0[i >> 1] = dg;
                const float gval = cfa[i] + dg;
                rgbgreen[i] = gval;
                const bool ny = nyquist2[i >> 1] != 0;
                Dgrb2[2 * (i >> 1)] = ny ? sqr(gval - xdiv2f(rgbgreen[i - 1] + rgbgreen[i + 1])) : 0.f;
                Dgrb2[2 * (i >> 1) + 1] = ny ? sqr(gval - xdiv2f(rgbgreen[i - v1] + rgbgreen[i + v1])) : 0.f;
            }
        }
        }
        SYNC_AFTER(11);

        // ---- P10: refine nyquist areas with the G curvature (L979-999) ----
        if constexpr (RUN(12))
        if (doNyquist) {
            const float gq0 = 0.169917f, gq1 = 0.108947f, gq2 = 0.069855f, gq3 = 0.0287182f;
            // two sub-steps: the reference updates Dgrb0/rgbgreen of the site itself only and
            // reads Dgrb2 (not written here), so one parallel pass is exact.
            FOR_ITEMS(nystartrow, nyendrow, ngroups(nystartcol, nyendcol, 2)) {
                const int cc = nystartcol + (fc(filters, rr, 2) & 1) + 2 * it;
                const int i = rr * ts + cc;
                if (cc < nyendcol && nyquist2[i >> 1]) {
#define DH(j) Dgrb2[2 * ((j) >> 1)]
#define DV(j) Dgrb2[2 * ((j) >> 1) + 1]
                    const float gvarh = epssq + (gq0 * DH(i) +
                                                 gq1 * (DH(i - m1) + DH(i + p1) + DH(i - p1) + DH(i + m1)) +
                                                 gq2 * (DH(i - v2) + DH(i - 2) + DH(i + 2) + DH(i + v2)) +
                                                 gq3 * (DH(i - m2) + DH(i + p2) + DH(i - p2) + DH(i + m2)));
                    const float gvarv = epssq + (gq0 * DV(i) +
                                                 gq1 * (DV(i - m1) + DV(i + p1) + DV(i - p1) + DV(i + m1)) +
                                                 gq2 * (DV(i - v2) + DV(i - 2) + DV(i + 2) + DV(i + v2)) +
                                                 gq3 * (DV(i - m2) + DV(i + p2) + DV(i - p2) + DV(i + m2)));
#undef DH
#undef DV
                    const float dg = (hcd[i] * gvarv + vcd[i] * gvarh) / (gvarv + gvarh);
                    Dgrb0[i >> 1] = dg;
                    rgbgreen[i] = cfa[i] + dg;
                }
            }
        }
        SYNC_AFTER(12);

        // ---- P11: diagonal gradients (L1004-1027); delp/delm overwrite cddiffsq/nyquist2 ----
        if constexpr (RUN(13))
        FOR_ITEMS(6, rr1 - 6, 4 * ngroups(6, cc1 - 6, 8)) {
            const int i = rr * ts + 6 + 2 * it;
            const bool rbEven = (fc(filters, rr, 2) & 1) == 0;
            const int g = rbEven ? i + 1 : i; // green site of the pair
            const int q = rbEven ? i : i + 1; // red/blue site of the pair
            const float t = cfa[g];
            const float sp = sqr(t - cfa[g - p1]) + sqr(t - cfa[g + p1]);
            const float sm = sqr(t - cfa[g - m1]) + sqr(t - cfa[g + m1]);
            delp[i >> 1] = fabsf(cfa[q + p1] - cfa[q - p1]);
            delm[i >> 1] = fabsf(cfa[q + m1] - cfa[q - m1]);
            Dgrbsq1m[i >> 1] = sm;
            Dgrbsq1p[i >> 1] = sp;
        }
        SYNC_AFTER(13);

        // ---- P12: diagonal interpolation of R+B and plus/minus weight (L1061-1121) ----
        if constexpr (RUN(14))
        FOR_ITEMS(8, rr1 - 8, 4 * ngroups(8, cc1 - 8, 8)) {
            const int par = fc(filters, rr, 2) & 1;
            if (it < 4 * ngroups(8 + par, cc1 - 8, 8)) {
                const int i = rr * ts + 8 + par + 2 * it, i1 = i >> 1;
                const float ge0 = 0.13719494435797422f, ge1 = 0.05640252782101291f;
                const float cfav = cfa[i];
                const float cse = cfa[i + m1], cnw = cfa[i - m1], cne = cfa[i + p1], csw = cfa[i - p1];
                const float rbse = rb_ratio(cfav, cse, cfa[i + m2]);
                const float rbnw = rb_ratio(cfav, cnw, cfa[i - m2]);
                float t1 = eps + delm[i1];
                const float wtse = t1 + delm[(i + m1) >> 1] + delm[(i + m2) >> 1];
                const float wtnw = t1 + delm[(i - m1) >> 1] + delm[(i - m2) >> 1];
                const float rbmv = (wtse * rbnw + wtnw * rbse) / (wtse + wtnw);
                const float rbm_out = rb_bound(rbmv, cfav, cnw, cse, clip_pt);
                const float rbne = rb_ratio(cfav, cne, cfa[i + p2]);
                const float rbsw = rb_ratio(cfav, csw, cfa[i - p2]);
                t1 = eps + delp[i1];
                const float wtne = t1 + delp[(i + p1) >> 1] + delp[(i + p2) >> 1];
                const float wtsw = t1 + delp[(i - p1) >> 1] + delp[(i - p2) >> 1];
                const float rbpv = (wtne * rbsw + wtsw * rbne) / (wtne + wtsw);
                const float rbp_out = rb_bound(rbpv, cfav, csw, cne, clip_pt);
                const float *Dm = Dgrbsq1m, *Dp = Dgrbsq1p;
                const float rbvarm = epssq + (ge0 * (Dm[(i - v1) >> 1] + Dm[(i - 1) >> 1] + Dm[(i + 1) >> 1] + Dm[(i + v1) >> 1]) +
                                              ge1 * (Dm[(i - v2 - 1) >> 1] + Dm[(i - v2 + 1) >> 1] + Dm[(i - 2 - v1) >> 1] + Dm[(i + 2 - v1) >> 1] +
                                                     Dm[(i - 2 + v1) >> 1] + Dm[(i + 2 + v1) >> 1] + Dm[(i + v2 - 1) >> 1] + Dm[(i + v2 + 1) >> 1]));
                const float rbvarp = epssq + (ge0 * (Dp[(i - v1) >> 1] + Dp[(i - 1) >> 1] + Dp[(i + 1) >> 1] + Dp[(i + v1) >> 1]) +
                                              ge1 * (Dp[(i - v2 - 1) >> 1] + Dp[(i - v2 + 1) >> 1] + Dp[(i - 2 - v1) >> 1] + Dp[(i + 2 - v1) >> 1] +
                                                     Dp[(i - 2 + v1) >> 1] + Dp[(i + 2 + v1) >> 1] + Dp[(i + v2 - 1) >> 1] + Dp[(i + v2 + 1) >> 1]));
                rbm[i1] = rbm_out;
                rbp[i1] = rbp_out;
                pmwt[i1] = rbvarm / (rbvarp + rbvarm);
            }
        }
        SYNC_AFTER(14);

        // ---- P13: pmwt refined in place row by row, rbint (L1213-1223); same scheme as P9 ----
        if constexpr (RUN(15)) {
        for (int t = tid; t < ts * tsh; t += NTT) s_plane[t] = pmwt[t];
        __syncthreads();
        if (tid < 64) {
            for (int rr = 10; rr < rr1 - 10; ++rr) {
                const int par = fc(filters, rr, 2) & 1;
                const int nit = 4 * ngroups(10 + par, cc1 - 10, 8);
                for (int j = tid; j < nit; j += 64) {
                    const int i = rr * ts + 10 + par + 2 * j, i1 = i >> 1;
                    const float alt = 0.25f * (s_plane[(i - m1) >> 1] + s_plane[(i + p1) >> 1] + s_plane[(i - p1) >> 1] + s_plane[(i + m1) >> 1]);
                    const float t = s_plane[i1];
                    s_plane[i1] = fabsf(0.5f - t) < fabsf(0.5f - alt) ? alt : t;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        FOR_ITEMS(10, rr1 - 10, 72) {
            const int par = fc(filters, rr, 2) & 1;
            if (it < 4 * ngroups(10 + par, cc1 - 10, 8)) {
                const int i = rr * ts + 10 + par + 2 * it, i1 = i >> 1;
                const float t = s_plane[i1];
                pmwt[i1] = t;
                rbint[i1] = 0.5f * (cfa[i] + intp(t, rbp[i1], rbm[i1]));
            }
        }
        }
        SYNC_AFTER(15);

        // ---- P14: G re-interpolated from R+B where the diagonal weight is more decisive (L1241-1297) ----
        if constexpr (RUN(16))
        FOR_ITEMS(12, rr1 - 12, 4 * ngroups(12, cc1 - 12, 8)) {
            const int par = fc(filters, rr, 2) & 1;
            if (it < 4 * ngroups(12 + par, cc1 - 12, 8)) {
                const int i = rr * ts + 12 + par + 2 * it, i1 = i >> 1;
                const float hw = hvwt[i1];
                if (fabsf(0.5f - pmwt[i1]) >= fabsf(0.5f - hw)) {
                    const float rb = rbint[i1];
                    const float cu = cfa[i - v1], cd = cfa[i + v1], cl = cfa[i - 1], cr = cfa[i + 1];
                    const float gu = g_dir(rb, cu, rbint[i1 - v1]);
                    const float gd = g_dir(rb, cd, rbint[i1 + v1]);
                    const float d0u = dirwts0[i - v1], d0d = dirwts0[i + v1];
                    float Gintv = (d0u * gd + d0d * gu) / (d0d + d0u);
                    Gintv = g_bound(Gintv, rb, cu, cd, clip_pt);
                    const float gl = g_dir(rb, cl, rbint[i1 - 1]);
                    const float gr = g_dir(rb, cr, rbint[i1 + 1]);
                    const float d1l = dirwts1[i - 1], d1r = dirwts1[i + 1];
                    float Ginth = (d1l * gr + d1r * gl) / (d1l + d1r);
                    Ginth = g_bound(Ginth, rb, cl, cr, clip_pt);
                    const float gval = intp(hw, Gintv, Ginth);
                    rgbgreen[i] = gval;
                    Dgrb0[i1] = gval - cfa[i];
                }
            }
        }
        SYNC_AFTER(16);

        // ---- P15: split G-B out of G-R on the B rows (L1381-1386) ----
        if constexpr (RUN(17)) {
            const int r0 = 13 - ey;
            const int nrow = rr1 - 12 > r0 ? (rr1 - 12 - r0 + 1) / 2 : 0;
            for (int t = tid; t < nrow * tsh; t += NTT) {
                const int rr = r0 + 2 * (t / tsh), k = t % tsh;
                const int i1 = ((rr * ts + 13 - ex) >> 1) + k;
                if (i1 < ((rr * ts + cc1 - 12) >> 1)) {
                    Dgrb1[i1] = Dgrb0[i1];
                    Dgrb0[i1] = 0.f;
                }
            }
        }
        SYNC_AFTER(17);

        // ---- P16: chrominance at the opposite-colour sites (L1394-1408) ----
        if constexpr (RUN(18))
        FOR_ITEMS(14, rr1 - 14, 4 * ngroups(14, cc1 - 14, 8)) {
            const int par = fc(filters, rr, 2) & 1;
            if (it < 4 * ngroups(14 + par, cc1 - 14, 8)) {
                const int cc0 = 14 + par;
                const int c = 1 - (int)fc(filters, rr, cc0) / 2;
                float *D = c ? Dgrb1 : Dgrb0;
                const int i = rr * ts + cc0 + 2 * it;
#define DG(j) D[(j) >> 1]
                const float dnw = DG(i - m1), dse = DG(i + m1), dne = DG(i + p1), dsw = DG(i - p1);
                const float dnw3 = DG(i - m3), dse3 = DG(i + m3), dne3 = DG(i + p3), dsw3 = DG(i - p3);
                const float temp = eps + fabsf(dnw - dse);
                const float temp2 = eps + fabsf(dne - dsw);
                const float wtnw = 1.f / (temp + fabsf(dnw - dnw3) + fabsf(dse - dnw3));
                const float wtne = 1.f / (temp2 + fabsf(dne - dne3) + fabsf(dsw - dne3));
                const float wtsw = 1.f / (temp2 + fabsf(dsw - dse3) + fabsf(dne - dsw3));
                const float wtse = 1.f / (temp + fabsf(dse - dsw3) + fabsf(dnw - dse3));
                const float val = (wtnw * (1.325f * dnw - 0.175f * dnw3 - 0.075f * (DG(i - m1 - 2) + DG(i - m1 - v2))) +
                                   wtne * (1.325f * dne - 0.175f * dne3 - 0.075f * (DG(i + p1 + 2) + DG(i + p1 + v2))) +
                                   wtsw * (1.325f * dsw - 0.175f * dsw3 - 0.075f * (DG(i - p1 - 2) + DG(i - p1 - v2))) +
                                   wtse * (1.325f * dse - 0.175f * dse3 - 0.075f * (DG(i + m1 + 2) + DG(i + m1 + v2)))) /
                                  (wtnw + wtne + wtsw + wtse);
#undef DG
                D[i >> 1] = val;
            }
        }
        SYNC_AFTER(18);

        // ---- P17/P18: write R, G, B for [16,rr1-16) x [16,cc1-16) (L1441-1565) ----
        if constexpr (RUN(19))
        FOR_ITEMS(16, rr1 - 16, cc1 - 32 > 0 ? cc1 - 32 : 0) {
            const int cc = 16 + it, i = rr * ts + cc;
            const float gval = rgbgreen[i];
            float r, b;
            if (fc(filters, rr, cc) & 1) {
                const float h_up = hvwt[(i - v1) >> 1], h_dn = hvwt[(i + v1) >> 1], h_r = hvwt[(i + 1) >> 1], h_l = hvwt[(i - 1) >> 1];
                const float temp = 1.f / (h_up + 2.f - h_r - h_l + h_dn);
                r = gval - (h_up * Dgrb0[(i - v1) >> 1] + (1.f - h_r) * Dgrb0[(i + 1) >> 1] + (1.f - h_l) * Dgrb0[(i - 1) >> 1] + h_dn * Dgrb0[(i + v1) >> 1]) * temp;
                b = gval - (h_up * Dgrb1[(i - v1) >> 1] + (1.f - h_r) * Dgrb1[(i + 1) >> 1] + (1.f - h_l) * Dgrb1[(i - 1) >> 1] + h_dn * Dgrb1[(i + v1) >> 1]) * temp;
            } else {
                r = gval - Dgrb0[i >> 1];
                b = gval - Dgrb1[i >> 1];
            }
            const size_t o = (size_t)(rr + top) * a.out_stride + (left + cc);
            a.red[o] = sse_max(65535.f * r, 0.f);
            a.blue[o] = sse_max(65535.f * b, 0.f);
            a.green[o] = sse_max(gval * 65535.f, 0.f);
        }
        if constexpr (G == 1) __syncthreads();   // fused kernel: the workgroup's next tile reuses the arena
    }
#undef RUN
#undef SYNC_AFTER
}

namespace {
template <int K, int G>
void launch_phase(const AmazeArgs &a, hipStream_t stream)
{
    hipLaunchKernelGGL((amaze_kernel<K, K, G, 1>), dim3(a.ntiles * G), dim3(AMAZE_THREADS), 0, stream, a);
}
} // namespace

hipError_t launch_amaze(const AmazeArgs &a, int grid, hipStream_t stream)
{
    // Default: the fused kernel (one workgroup walks a tile through all phases; 6.9 ms at 45 MP).  ARTGPU_AMAZE_SPLIT=1 launches
    // one kernel per phase over all tiles instead (needs one arena per tile): bit-identical, 7.5 ms -- every phase then streams
    // its planes of ALL tiles through HBM, which makes it the per-phase bandwidth profile of the algorithm
    // (profiles/r1/amaze_split_phase_stats.csv) rather than the fast path.
    const bool split = a.split && !a.tile_list && grid >= a.ntiles;
    if (!split) {
        hipLaunchKernelGGL((amaze_kernel<0, AMAZE_NPHASES - 1, 1, AMAZE_MIN_WAVES>), dim3(grid), dim3(AMAZE_THREADS), 0, stream, a);
        return hipGetLastError();
    }
    launch_phase<0, 1>(a, stream);    // clear + tile initialisation (ordered sub-steps)
    launch_phase<1, 2>(a, stream);    // P1 gradients
    launch_phase<2, 2>(a, stream);    // P2 colour differences
    launch_phase<3, 2>(a, stream);    // P3a
    launch_phase<4, 1>(a, stream);    // P3b + the vcd column walk
    launch_phase<5, 2>(a, stream);    // P3d
    launch_phase<6, 2>(a, stream);    // P4 + P5
    launch_phase<7, 2>(a, stream);    // P6 flags + bounding box
    launch_phase<8, 1>(a, stream);    // nyquist2 clear
    launch_phase<9, 1>(a, stream);    // P7 vote
    launch_phase<10, 1>(a, stream);   // P8 area interpolation (LDS site list)
    launch_phase<11, 1>(a, stream);   // P9 hvwt row recurrence (LDS plane) + consumers
    launch_phase<12, 2>(a, stream);   // P10
    launch_phase<13, 2>(a, stream);   // P11
    launch_phase<14, 2>(a, stream);   // P12
    launch_phase<15, 1>(a, stream);   // P13 pmwt row recurrence (LDS plane) + consumers
    launch_phase<16, 2>(a, stream);   // P14
    launch_phase<17, 1>(a, stream);   // P15
    launch_phase<18, 2>(a, stream);   // P16
    launch_phase<19, 2>(a, stream);   // output
    return hipGetLastError();
}

} // namespace artgpu
