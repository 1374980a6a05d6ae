// art_amd/csrc/rcd_stream_core.h -- RCD demosaic with the tile state in LDS (row streaming).
//
// Replaces RawImageSource::rcd_demosaic (reference: rtengine/rcd_demosaic.cc:51-347) tile by tile: 194x194 reference tiles,
// stride 176, write-back margin 9 -- the tile grid is part of the result, because VH_Dir / PQ_Dir / the interpolated planes
// are only defined on [4, n-4) of each tile and what is read outside is the cleared buffer (0) or, for PQ_Dir, the lpf values
// it aliases (L101-103,199).  RCD is a pure stencil chain (no row recurrences), so a workgroup walks a tile top to bottom
// R rows per iteration; every step of the reference runs a fixed number of rows behind the load front on a ring buffer:
//
//   front A = rows [0, A) of the tile are in the cfa ring.  One iteration (A += R), four LDS barriers:
//     I1: issue the global loads of rows [A, A+R) | step 2 lpf @1 | step 4.0 P/Q hpf @3 | step 1 VH_Dir @4
//     I2: step 3 green at R/B @5 | step 4.1 PQ_Dir @4
//     I3: step 4.2 R/B at B/R @7
//     I4: step 4.3 R/B at green + write-back of the row @10 | the loaded rows go to the cfa ring
//   a stage "@L" handles rows [A-L-R, A-L).  What a stage reads was written in an earlier interval.
//
// Layout: full-resolution rows are stored de-interleaved (even columns | odd columns, 97 + 97 floats), half-resolution planes
// with the reference's own half index (row * 97 + col / 2), so every access of a wave is unit-stride (no bank conflicts) and the
// column offsets of a stencil are immediates once the parity of the site's column is a template parameter.
// Positions a step does not compute hold what the reference's cleared buffer holds (0; lpf for PQ_Dir): every stage writes its
// whole ring row, value or default, so readers need no domain checks.
//
// Work distribution: an item is (row j of the batch, half-column slot t in 0..95): three waves share two rows (j, j+2) of equal
// colour layout -- lanes 0..63 of each row, and one wave with slots 64..95 of both (lane >= 32: the second row).  Rows are
// wave-uniform (or a two-way select), so ring addressing is scalar arithmetic.
//
// The same source compiles for the device (rcd_stream.hip) and, with RCS_EMUL, as a sequential CPU emulation
// (tests/emul/rcd_stream_emul.cc, test harness only) that checks the schedule against the oracle without a GPU.
#pragma once

#ifdef RCS_EMUL
#include <math.h>
#include <stdint.h>
#include <string.h>
#define RCS_DEV static inline
#define RCS_MEM inline
typedef float *rcs_lf;
typedef const float *rcs_gcf;
typedef float *rcs_gf;
namespace rcs {
static inline unsigned fc(unsigned filters, unsigned row, unsigned col) { return (filters >> (((((row) << 1) & 14u) + ((col) & 1u)) << 1)) & 3u; }
static inline float std_min(float a, float b) { return b < a ? b : a; }
static inline float std_max(float a, float b) { return a < b ? b : a; }
static inline float sqr(float x) { return x * x; }
static inline float intp(float a, float b, float c) { return a * b + (1.f - a) * c; }
static inline float lim01(float a) { return std_max(0.f, std_min(a, 1.f)); }
// ring-slot tags: every consumed read has to find the row it expects (ring depths, stage lags)
extern int *g_tag;
extern long long g_tag_errors, g_tag_first[4];
extern int g_seq;
static inline float ld_checked(const float *lds, int i, int row, int line)
{
    const int want = g_seq * 4096 + row + 64;
    if (g_tag[i] != want && !g_tag_errors++) { g_tag_first[0] = line; g_tag_first[1] = row; g_tag_first[2] = g_tag[i]; g_tag_first[3] = i; }
    return lds[i];
}
static inline void st_tagged(float *lds, int i, int row, float v) { lds[i] = v; g_tag[i] = g_seq * 4096 + row + 64; }
}
#define RCS_LD(i, row) rcs::ld_checked(lds, (i), (row), __LINE__)
#define RCS_UNIFORM(x) (x)
#define RCS_ST(i, row, v) rcs::st_tagged(lds, (i), (row), (v))
#else
#include <hip/hip_runtime.h>
#include "devmath.h"
#define RCS_DEV __device__ __forceinline__
#define RCS_MEM __device__ __forceinline__
typedef __attribute__((address_space(3))) float *rcs_lf;
typedef const __attribute__((address_space(1))) float *rcs_gcf;
typedef __attribute__((address_space(1))) float *rcs_gf;
namespace rcs {
using artgpu::fc; using artgpu::std_min; using artgpu::std_max; using artgpu::sqr; using artgpu::intp; using artgpu::lim01;
}
#define RCS_LD(i, row) lds[(i)]
namespace rcs { __device__ __forceinline__ int opaque_sgpr(int x) { asm("" : "+s"(x)); return x; } }
#define RCS_UNIFORM(x) rcs::opaque_sgpr(__builtin_amdgcn_readfirstlane(x))
#define RCS_ST(i, row, v) lds[(i)] = (v)
#endif

namespace rcs {

constexpr int TS = 194, HW = 97, BORDER = 9, TSN = TS - 2 * BORDER;
constexpr float eps = 1e-5f, epssq = 1e-10f, scale = 65536.f;
// rows behind the front
constexpr int L_S2 = 1, L_S40 = 3, L_S1 = 4, L_S3 = 5, L_S41 = 4, L_S42 = 7, L_S43 = 10;

// A ring: D logical rows (row r lives in slot r mod D) + E mirror rows (slot s < E is also kept at s + D), so that the rows
// r+kmin .. r+kmax a stencil reads (kmax - kmin <= E) are CONTIGUOUS from slot (r+kmin) mod D: one base address per stage and
// ring, every row / column offset of the stencil an immediate.  D = newest row written while the oldest is still read + 1 (the
// emulation's row tags check it under adversarial thread orders).
template <int R>
struct Cfg {
    static_assert(R % 4 == 0, "two rows of one colour layout per wave triple");
    static constexpr int NW = 3 * R / 2, NT = NW * 64;
    static constexpr int D_CFA = 2 * R + 13, D_VH = R + 7, D_LPF = R + 6, D_PQH = R + 2, D_PQD = R + 4, D_G = R + 6, D_RBD = R + 6;
    static constexpr int E_CFA = 8, E_VH = 2, E_LPF = 4, E_PQH = 2, E_PQD = 2, E_G = 4, E_RBD = 6;
    static constexpr int O_CFA = 8;                                // a few floats of slack in front: slot -1 of ring row 0 is addressed, never consumed
    static constexpr int O_VH = O_CFA + (D_CFA + E_CFA) * TS;
    static constexpr int O_LPF = O_VH + (D_VH + E_VH) * TS;
    static constexpr int O_PQH = O_LPF + (D_LPF + E_LPF) * HW;     // [P | Q] per row
    static constexpr int O_PQD = O_PQH + (D_PQH + E_PQH) * TS;
    static constexpr int O_G = O_PQD + (D_PQD + E_PQD) * HW;       // green at red / blue sites
    static constexpr int O_RBD = O_G + (D_G + E_G) * HW;           // red at blue / blue at red sites
    static constexpr int LDS_FLOATS = O_RBD + (D_RBD + E_RBD) * HW + 8;
};

struct Tile {
    rcs_gcf raw;            // tile origin
    long rs;
    rcs_gf red, green, blue;
    long os;
    int rows, cols;         // tileRows, tilecols (<= 194)
    unsigned filters;       // the tile origin is even in both directions: FC(local) == FC(global)
    int vec2;               // out rows are 8-byte aligned at even columns: pairs go out as one store
};

template <int D>
RCS_DEV int wrap(int x) { return (int)((unsigned)(x + 4 * D) % (unsigned)D); }

// the rows of a thread: wave-uniform `a`, or (waves that carry slots 64..95 of two rows) a two-way select
template <bool SPLIT>
struct Rows {
    int a, b;
    bool hi;
    RCS_MEM int row() const
    {
        if constexpr (SPLIT) return hi ? b : a;
        else return a;
    }
    // ring slot of row + k
    // (forced into scalar registers: left alone, the compiler folds the select into the arithmetic and does the modulo per lane)
    template <int D>
    RCS_MEM int slot(int k) const
    {
        const int wa = RCS_UNIFORM(wrap<D>(a + k));
        if constexpr (SPLIT) {
            const int wb = RCS_UNIFORM(wrap<D>(b + k));
            return hi ? wb : wa;
        } else return wa;
    }
    // float offset of ring row (row + k) of a ring at `base` with row stride S
    template <int D, int S>
    RCS_MEM int off(int base, int k) const
    {
        const int oa = RCS_UNIFORM(base + wrap<D>(a + k) * S);
        if constexpr (SPLIT) {
            const int ob = RCS_UNIFORM(base + wrap<D>(b + k) * S);
            return hi ? ob : oa;
        } else return oa;
    }
};

// base (float index) of the thread's row + kmin in a ring: rows up to + E further are contiguous behind it
#define RCS_RO(RING, S, kmin) rw.template off<C::D_##RING, S>(C::O_##RING, (kmin))
// store into the thread's own row (and its mirror): `w` = rw.slot<D>(0)
#define RCS_PUT(RING, S, w, col, v)                                                         \
    do {                                                                                    \
        const int _i = C::O_##RING + (w) * (S) + (col);                                     \
        RCS_ST(_i, r, (v));                                                                 \
        if ((w) < C::E_##RING) RCS_ST(_i + C::D_##RING * (S), r, (v));                      \
    } while (0)
// element (row base o, column 2*t + par + d) of a de-interleaved full-resolution row; par and d are compile-time
#define RCS_FR(o, par, t, d) ((o) + ((((par) + (d)) & 1) * rcs::HW) + (t) + (((par) + (d)) >> 1))

RCS_DEV int png_of(unsigned filters, int r) { return (int)(fc(filters, (unsigned)(r & 1), 0u) & 1u); }   // parity of the row's red / blue columns

// ---- step 1 (L135-166): VH_Dir of columns 2t, 2t+1 ----
RCS_DEV float hpf7(float m3, float m2, float m1, float c0, float p1, float p2, float p3)
{
    return sqr((m3 - m1 - p1 + p3) - 3.f * (m2 + p2) + 6.f * c0);
}
template <class C, bool SPLIT>
RCS_DEV void st_s1(rcs_lf lds, const Tile &tl, const Rows<SPLIT> rw, int t)
{
    const int r = rw.row();
    if (r < 0 || r >= tl.rows) return;
    float v0 = 0.f, v1 = 0.f;
    if (r >= 4 && r < tl.rows - 4 && t >= 2 && 2 * t < tl.cols - 4) {
        const int bc = RCS_RO(CFA, TS, -4) + t;
        float e[9], o[9];      // columns 2t / 2t+1, rows r-4 .. r+4
#pragma unroll
        for (int k = -4; k <= 4; ++k) {
            e[k + 4] = RCS_LD(bc + (k + 4) * TS, r + k);
            o[k + 4] = RCS_LD(bc + (k + 4) * TS + HW, r + k);
        }
        float he[5], ho[5];    // row r: even columns 2t-4 .. 2t+4, odd columns 2t-3 .. 2t+5
#pragma unroll
        for (int d = -2; d <= 2; ++d) {
            he[d + 2] = d == 0 ? e[4] : RCS_LD(bc + 4 * TS + d, r);
            ho[d + 2] = d == 0 ? o[4] : RCS_LD(bc + 4 * TS + HW + d, r);
        }
        // horizontal high-pass at columns 2t-1, 2t, 2t+1, 2t+2
        const float hm1 = hpf7(he[0], ho[0], he[1], ho[1], he[2], ho[2], he[3]);
        const float h0 = hpf7(ho[0], he[1], ho[1], he[2], ho[2], he[3], ho[3]);
        const float h1 = hpf7(he[1], ho[1], he[2], ho[2], he[3], ho[3], he[4]);
        const float h2 = hpf7(ho[1], he[2], ho[2], he[3], ho[3], he[4], ho[4]);
        const float Ve = std_max(epssq, hpf7(e[0], e[1], e[2], e[3], e[4], e[5], e[6]) + hpf7(e[1], e[2], e[3], e[4], e[5], e[6], e[7]) + hpf7(e[2], e[3], e[4], e[5], e[6], e[7], e[8]));
        const float Vo = std_max(epssq, hpf7(o[0], o[1], o[2], o[3], o[4], o[5], o[6]) + hpf7(o[1], o[2], o[3], o[4], o[5], o[6], o[7]) + hpf7(o[2], o[3], o[4], o[5], o[6], o[7], o[8]));
        const float He = std_max(epssq, hm1 + h0 + h1);
        const float Ho = std_max(epssq, h0 + h1 + h2);
        v0 = Ve / (Ve + He);
        v1 = 2 * t + 1 < tl.cols - 4 ? Vo / (Vo + Ho) : 0.f;
    }
    const int w = rw.template slot<C::D_VH>(0);
    RCS_PUT(VH, TS, w, t, v0);
    RCS_PUT(VH, TS, w, HW + t, v1);
}

// ---- step 2 (L169-175): low-pass at the red / blue site 2t + P ----
template <class C, int P, bool SPLIT>
RCS_DEV void st_s2(rcs_lf lds, const Tile &tl, const Rows<SPLIT> rw, int t)
{
    const int r = rw.row();
    if (r < 0 || r >= tl.rows) return;
    float v = 0.f;
    if (r >= 2 && r < tl.rows - 2 && t >= 1 && 2 * t + P < tl.cols - 2) {
        const int om = RCS_RO(CFA, TS, -1), o0 = om + TS, op = om + 2 * TS;
        v = RCS_LD(RCS_FR(o0, P, t, 0), r) +
            0.5f * (RCS_LD(RCS_FR(om, P, t, 0), r - 1) + RCS_LD(RCS_FR(op, P, t, 0), r + 1) + RCS_LD(RCS_FR(o0, P, t, -1), r) + RCS_LD(RCS_FR(o0, P, t, 1), r)) +
            0.25f * (RCS_LD(RCS_FR(om, P, t, -1), r - 1) + RCS_LD(RCS_FR(om, P, t, 1), r - 1) + RCS_LD(RCS_FR(op, P, t, -1), r + 1) + RCS_LD(RCS_FR(op, P, t, 1), r + 1));
    }
    const int w = rw.template slot<C::D_LPF>(0);
    RCS_PUT(LPF, HW, w, t, v);
}

// ---- step 4.0 (L213-218): diagonal high-pass at the odd column 2t + 1 of every row ----
template <class C, bool SPLIT>
RCS_DEV void st_s40(rcs_lf lds, const Tile &tl, const Rows<SPLIT> rw, int t)
{
    const int r = rw.row();
    if (r < 0 || r >= tl.rows) return;
    float p = 0.f, q = 0.f;
    if (r >= 3 && r < tl.rows - 3 && t >= 1 && 2 * t + 1 < tl.cols - 3) {
        const int bc = RCS_RO(CFA, TS, -3);
#define CF(k, d) RCS_LD(RCS_FR(bc + ((k) + 3) * TS, 1, t, d), r + (k))
        const float c0 = CF(0, 0);
        p = sqr((CF(-3, -3) - CF(-1, -1) - CF(1, 1) + CF(3, 3)) - 3.f * (CF(-2, -2) + CF(2, 2)) + 6.f * c0);
        q = sqr((CF(-3, 3) - CF(-1, 1) - CF(1, -1) + CF(3, -3)) - 3.f * (CF(-2, 2) + CF(2, -2)) + 6.f * c0);
#undef CF
    }
    const int w = rw.template slot<C::D_PQH>(0);
    RCS_PUT(PQH, TS, w, t, p);
    RCS_PUT(PQH, TS, w, HW + t, q);
}

// ---- step 3 (L178-206): green at the red / blue site 2t + P ----
template <class C, int P, bool SPLIT>
RCS_DEV void st_s3(rcs_lf lds, const Tile &tl, const Rows<SPLIT> rw, int t)
{
    const int r = rw.row();
    if (r < 0 || r >= tl.rows) return;
    float g = 0.f;
    if (r >= 4 && r < tl.rows - 4 && t >= 2 && 2 * t + P < tl.cols - 4) {
        const int bc = RCS_RO(CFA, TS, -4);
#define CF(k, d) RCS_LD(RCS_FR(bc + ((k) + 4) * TS, P, t, d), r + (k))
        const float cfai = CF(0, 0);
        const float cN1 = CF(-1, 0), cN2 = CF(-2, 0), cN3 = CF(-3, 0), cN4 = CF(-4, 0);
        const float cS1 = CF(1, 0), cS2 = CF(2, 0), cS3 = CF(3, 0), cS4 = CF(4, 0);
        const float cW1 = CF(0, -1), cW2 = CF(0, -2), cW3 = CF(0, -3), cW4 = CF(0, -4);
        const float cE1 = CF(0, 1), cE2 = CF(0, 2), cE3 = CF(0, 3), cE4 = CF(0, 4);
#undef CF
        const float N_Grad = eps + (fabsf(cN1 - cS1) + fabsf(cfai - cN2)) + (fabsf(cN1 - cN3) + fabsf(cN2 - cN4));
        const float S_Grad = eps + (fabsf(cN1 - cS1) + fabsf(cfai - cS2)) + (fabsf(cS1 - cS3) + fabsf(cS2 - cS4));
        const float W_Grad = eps + (fabsf(cW1 - cE1) + fabsf(cfai - cW2)) + (fabsf(cW1 - cW3) + fabsf(cW2 - cW4));
        const float E_Grad = eps + (fabsf(cW1 - cE1) + fabsf(cfai - cE2)) + (fabsf(cE1 - cE3) + fabsf(cE2 - cE4));
        // lpf[lp -+ w1] of the reference is TWO half-rows away: the same-colour site two rows up / down
        const int bl = RCS_RO(LPF, HW, -2) + t;
        const float lpfi = RCS_LD(bl + 2 * HW, r);
        const float lN = RCS_LD(bl, r - 2), lS = RCS_LD(bl + 4 * HW, r + 2);
        const float lW = RCS_LD(bl + 2 * HW - 1, r), lE = RCS_LD(bl + 2 * HW + 1, r);
        const float N_Est = cN1 * (lpfi + lpfi) / (eps + lpfi + lN);
        const float S_Est = cS1 * (lpfi + lpfi) / (eps + lpfi + lS);
        const float W_Est = cW1 * (lpfi + lpfi) / (eps + lpfi + lW);
        const float E_Est = cE1 * (lpfi + lpfi) / (eps + lpfi + lE);
        const float V_Est = (S_Grad * N_Est + N_Grad * S_Est) / (N_Grad + S_Grad);
        const float H_Est = (W_Grad * E_Est + E_Grad * W_Est) / (E_Grad + W_Grad);
        const int vm = RCS_RO(VH, TS, -1), v0 = vm + TS, vp = vm + 2 * TS;
        const float VH_C = RCS_LD(RCS_FR(v0, P, t, 0), r);
        const float VH_N = 0.25f * ((RCS_LD(RCS_FR(vm, P, t, -1), r - 1) + RCS_LD(RCS_FR(vm, P, t, 1), r - 1)) + (RCS_LD(RCS_FR(vp, P, t, -1), r + 1) + RCS_LD(RCS_FR(vp, P, t, 1), r + 1)));
        const float VH_Disc = fabsf(0.5f - VH_C) < fabsf(0.5f - VH_N) ? VH_N : VH_C;
        g = intp(VH_Disc, H_Est, V_Est);
    }
    const int w = rw.template slot<C::D_G>(0);
    RCS_PUT(G, HW, w, t, g);
}

// ---- step 4.1 (L221-227): PQ_Dir at the red / blue site 2t + P; elsewhere the lpf value it aliases (L103) ----
template <class C, int P, bool SPLIT>
RCS_DEV void st_s41(rcs_lf lds, const Tile &tl, const Rows<SPLIT> rw, int t)
{
    const int r = rw.row();
    if (r < 0 || r >= tl.rows) return;
    float v;
    if (r >= 4 && r < tl.rows - 4 && t >= 2 && 2 * t + P < tl.cols - 4) {
        constexpr int W = (P - 1) >> 1;          // slot of column c - 1 relative to t
        const int hm = RCS_RO(PQH, TS, -1) + t, h0 = hm + TS, hp = hm + 2 * TS;
        const float P_Stat = std_max(epssq, RCS_LD(hm + W, r - 1) + RCS_LD(h0, r) + RCS_LD(hp + W + 1, r + 1));
        const float Q_Stat = std_max(epssq, RCS_LD(hm + HW + W + 1, r - 1) + RCS_LD(h0 + HW, r) + RCS_LD(hp + HW + W, r + 1));
        v = P_Stat / (P_Stat + Q_Stat);
    } else {
        v = r >= 2 && r < tl.rows - 2 ? RCS_LD(RCS_RO(LPF, HW, 0) + t, r) : 0.f;
    }
    const int w = rw.template slot<C::D_PQD>(0);
    RCS_PUT(PQD, HW, w, t, v);
}

// ---- step 4.2 (L230-258): red at blue / blue at red, site 2t + P ----
template <class C, int P, bool SPLIT>
RCS_DEV void st_s42(rcs_lf lds, const Tile &tl, const Rows<SPLIT> rw, int t)
{
    const int r = rw.row();
    if (r < 0 || r >= tl.rows) return;
    float v = 0.f;
    if (r >= 4 && r < tl.rows - 4 && t >= 2 && 2 * t + P < tl.cols - 4) {
        constexpr int W = (P - 1) >> 1, E = (P + 1) >> 1;      // slots of columns c - 1, c + 1 relative to t
        const int qm = RCS_RO(PQD, HW, -1) + t, q0 = qm + HW, qp = qm + 2 * HW;
        const float PQ_C = RCS_LD(q0, r);
        const float PQ_N = 0.25f * (RCS_LD(qm + W, r - 1) + RCS_LD(qm + W + 1, r - 1) + RCS_LD(qp + W, r + 1) + RCS_LD(qp + W + 1, r + 1));
        const float PQ_Disc = (fabsf(0.5f - PQ_C) < fabsf(0.5f - PQ_N)) ? PQ_N : PQ_C;
        const int bc = RCS_RO(CFA, TS, -3);
#define CF(k, d) RCS_LD(RCS_FR(bc + ((k) + 3) * TS, P, t, d), r + (k))
        const float rNW = CF(-1, -1), rNE = CF(-1, 1), rSW = CF(1, -1), rSE = CF(1, 1);
        const float rNW3 = CF(-3, -3), rNE3 = CF(-3, 3), rSW3 = CF(3, -3), rSE3 = CF(3, 3);
#undef CF
        const int gm2 = RCS_RO(G, HW, -2) + t, gm1 = gm2 + HW, g0o = gm2 + 2 * HW, gp1 = gm2 + 3 * HW, gp2 = gm2 + 4 * HW;
        const float g0 = RCS_LD(g0o, r);
        const float NW_Grad = eps + fabsf(rNW - rSE) + fabsf(rNW - rNW3) + fabsf(g0 - RCS_LD(gm2 - 1, r - 2));
        const float NE_Grad = eps + fabsf(rNE - rSW) + fabsf(rNE - rNE3) + fabsf(g0 - RCS_LD(gm2 + 1, r - 2));
        const float SW_Grad = eps + fabsf(rNE - rSW) + fabsf(rSW - rSW3) + fabsf(g0 - RCS_LD(gp2 - 1, r + 2));
        const float SE_Grad = eps + fabsf(rNW - rSE) + fabsf(rSE - rSE3) + fabsf(g0 - RCS_LD(gp2 + 1, r + 2));
        const float NW_Est = rNW - RCS_LD(gm1 + W, r - 1);
        const float NE_Est = rNE - RCS_LD(gm1 + E, r - 1);
        const float SW_Est = rSW - RCS_LD(gp1 + W, r + 1);
        const float SE_Est = rSE - RCS_LD(gp1 + E, r + 1);
        const float P_Est = (NW_Grad * SE_Est + SE_Grad * NW_Est) / (NW_Grad + SE_Grad);
        const float Q_Est = (NE_Grad * SW_Est + SW_Grad * NE_Est) / (NE_Grad + SW_Grad);
        v = g0 + intp(PQ_Disc, Q_Est, P_Est);
    }
    const int w = rw.template slot<C::D_RBD>(0);
    RCS_PUT(RBD, HW, w, t, v);
    if (t == 95) RCS_PUT(RBD, HW, w, 96, 0.f);       // columns 192, 193 are outside every step's domain: step 4.3 at column 189 reads the cleared value
}

// ---- step 4.3 (L261-302): red and blue at the green site 2t + G, G = 1 - P; then the write-back of columns 2t, 2t+1 (L304-316) ----
template <class C, int G, bool SPLIT>
RCS_DEV void st_s43(rcs_lf lds, const Tile &tl, const Rows<SPLIT> rw, int t)
{
    const int r = rw.row();
    // only rows / columns that are written back: nothing reads this step's results
    if (r < BORDER || r >= tl.rows - BORDER) return;
    const int c0 = 2 * t;
    if (c0 + 1 < BORDER || c0 >= tl.cols - BORDER) return;
    float xh = 0.f, xv = 0.f;        // the row's own colour (left / right neighbours are native), the other one (up / down are native)
    const int bc = RCS_RO(CFA, TS, -3);
    const int bg = RCS_RO(G, HW, -1) + t;
    const int bb = RCS_RO(RBD, HW, -3) + t;
#define CF(k, d) RCS_LD(RCS_FR(bc + ((k) + 3) * TS, G, t, d), r + (k))
    if (2 * t + G < tl.cols - 4) {   // (r and the lower column bound are inside step 4.3's domain already)
        const int vm = RCS_RO(VH, TS, -1), v0 = vm + TS, vp = vm + 2 * TS;
        const float VH_C = RCS_LD(RCS_FR(v0, G, t, 0), r);
        const float VH_N = 0.25f * ((RCS_LD(RCS_FR(vm, G, t, -1), r - 1) + RCS_LD(RCS_FR(vm, G, t, 1), r - 1)) + (RCS_LD(RCS_FR(vp, G, t, -1), r + 1) + RCS_LD(RCS_FR(vp, G, t, 1), r + 1)));
        const float VH_Disc = (fabsf(0.5f - VH_C) < fabsf(0.5f - VH_N)) ? VH_N : VH_C;
        const float g0 = CF(0, 0);
        const float N1 = eps + fabsf(g0 - CF(-2, 0));
        const float S1 = eps + fabsf(g0 - CF(2, 0));
        const float W1 = eps + fabsf(g0 - CF(0, -2));
        const float E1 = eps + fabsf(g0 - CF(0, 2));
        constexpr int SW_ = (G - 1) >> 1, SE_ = (G + 1) >> 1, SW3 = (G - 3) >> 1, SE3 = (G + 3) >> 1;   // slots of columns c-1, c+1, c-3, c+3 relative to t (column c itself: t)
        const float gN = RCS_LD(bg, r - 1), gS = RCS_LD(bg + 2 * HW, r + 1), gW = RCS_LD(bg + HW + SW_, r), gE = RCS_LD(bg + HW + SE_, r);
        {   // the row's own colour: native left / right (cfa), interpolated by step 4.2 above / below
            const float rN = RCS_LD(bb + 2 * HW, r - 1), rS = RCS_LD(bb + 4 * HW, r + 1), rW = CF(0, -1), rE = CF(0, 1);
            const float SNabs = fabsf(rN - rS);
            const float EWabs = fabsf(rW - rE);
            const float N_Grad = N1 + SNabs + fabsf(rN - RCS_LD(bb, r - 3));
            const float S_Grad = S1 + SNabs + fabsf(rS - RCS_LD(bb + 6 * HW, r + 3));
            const float W_Grad = W1 + EWabs + fabsf(rW - CF(0, -3));
            const float E_Grad = E1 + EWabs + fabsf(rE - CF(0, 3));
            const float N_Est = rN - gN, S_Est = rS - gS, W_Est = rW - gW, E_Est = rE - gE;
            const float V_Est = (N_Grad * S_Est + S_Grad * N_Est) / (N_Grad + S_Grad);
            const float H_Est = (E_Grad * W_Est + W_Grad * E_Est) / (E_Grad + W_Grad);
            xh = g0 + intp(VH_Disc, H_Est, V_Est);
        }
        {   // the other colour: native above / below, interpolated left / right
            const int b0 = bb + 3 * HW;
            const float rN = CF(-1, 0), rS = CF(1, 0), rW = RCS_LD(b0 + SW_, r), rE = RCS_LD(b0 + SE_, r);
            const float SNabs = fabsf(rN - rS);
            const float EWabs = fabsf(rW - rE);
            const float N_Grad = N1 + SNabs + fabsf(rN - CF(-3, 0));
            const float S_Grad = S1 + SNabs + fabsf(rS - CF(3, 0));
            const float W_Grad = W1 + EWabs + fabsf(rW - RCS_LD(b0 + SW3, r));
            const float E_Grad = E1 + EWabs + fabsf(rE - RCS_LD(b0 + SE3, r));
            const float N_Est = rN - gN, S_Est = rS - gS, W_Est = rW - gW, E_Est = rE - gE;
            const float V_Est = (N_Grad * S_Est + S_Grad * N_Est) / (N_Grad + S_Grad);
            const float H_Est = (E_Grad * W_Est + W_Grad * E_Est) / (E_Grad + W_Grad);
            xv = g0 + intp(VH_Disc, H_Est, V_Est);
        }
    }
    // write-back: site A = the red / blue one (column 2t + p), site B = the green one
    constexpr int p = 1 - G;
    const bool row_red = fc(tl.filters, (unsigned)(r & 1), (unsigned)p) == 0;       // the row's red / blue sites are red
    const float cB = CF(0, 0), cA = CF(0, p - G);
#undef CF
    const float gi = RCS_LD(bg + HW, r), di = RCS_LD(bb + 3 * HW, r);
    const float rA = row_red ? cA : di, bA = row_red ? di : cA;
    const float rB = row_red ? xh : xv, bB = row_red ? xv : xh;
    const float re = std_max(0.f, (p ? rB : rA) * scale), ro = std_max(0.f, (p ? rA : rB) * scale);
    const float ge = std_max(0.f, (p ? cB : gi) * scale), go = std_max(0.f, (p ? gi : cB) * scale);
    const float be = std_max(0.f, (p ? bB : bA) * scale), bo = std_max(0.f, (p ? bA : bB) * scale);
    const long o = (long)r * tl.os + c0;
    const bool ve = c0 >= BORDER, vo = c0 + 1 < tl.cols - BORDER;
#ifndef RCS_EMUL
    if (tl.vec2 && ve && vo) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(1))) f2 *gf2;
        *(gf2)(tl.red + o) = f2{re, ro};
        *(gf2)(tl.green + o) = f2{ge, go};
        *(gf2)(tl.blue + o) = f2{be, bo};
        return;
    }
#endif
    if (ve) { tl.red[o] = re; tl.green[o] = ge; tl.blue[o] = be; }
    if (vo) { tl.red[o + 1] = ro; tl.green[o + 1] = go; tl.blue[o + 1] = bo; }
}

// ---- tile load (L126-131): columns 2t, 2t+1 (and 192, 193 with t == 95) of row r into registers, then into the cfa ring ----
struct LoadRegs { float e, o, e2, o2; };
template <bool SPLIT>
RCS_DEV void st_fetch(const Tile &tl, const Rows<SPLIT> rw, int t, LoadRegs &v)
{
    const int r = rw.row();
    v.e = v.o = v.e2 = v.o2 = 0.f;
    if (r < 0 || r >= tl.rows) return;
    rcs_gcf src = tl.raw + (long)r * tl.rs;
    if (2 * t < tl.cols) v.e = src[2 * t];
    if (2 * t + 1 < tl.cols) v.o = src[2 * t + 1];
    if (t == 95) {
        if (192 < tl.cols) v.e2 = src[192];
        if (193 < tl.cols) v.o2 = src[193];
    }
}
template <class C, bool SPLIT>
RCS_DEV void st_commit(rcs_lf lds, const Tile &tl, const Rows<SPLIT> rw, int t, const LoadRegs &v)
{
    const int r = rw.row();
    if (r < 0 || r >= tl.rows) return;
    const int w = rw.template slot<C::D_CFA>(0);
    RCS_PUT(CFA, TS, w, t, lim01(v.e / scale));
    RCS_PUT(CFA, TS, w, HW + t, lim01(v.o / scale));
    if (t == 95) {
        RCS_PUT(CFA, TS, w, 96, lim01(v.e2 / scale));
        RCS_PUT(CFA, TS, w, HW + 96, lim01(v.o2 / scale));
    }
}

// ---- the schedule: what one thread does in each of the four intervals of an iteration with front A ----
// `wave` 0 .. NW-1, `lane` 0 .. 63.  Wave triple q: rows j, j + 2 of the batch; the third wave of a triple carries slots 64..95 of both.
template <int R>
struct Sched {
    typedef Cfg<R> C;
    int ja, jb, t;
    bool split, hi;
    RCS_MEM Sched(int wave, int lane)
    {
        const int q = wave / 3, sub = wave - 3 * q;
        const int j = (q >> 1) * 4 + (q & 1);
        split = sub == 2;
        hi = split && lane >= 32;
        ja = sub == 1 ? j + 2 : j;
        jb = j + 2;
        t = split ? 64 + (lane & 31) : lane;
    }
    template <bool SPLIT>
    RCS_MEM Rows<SPLIT> rows(int base) const { return Rows<SPLIT>{base + ja, base + jb, hi}; }

#define RCS_CALL(fn, base, ...)                                            \
    do {                                                                   \
        if (split) fn<C, ##__VA_ARGS__, true>(lds, tl, rows<true>(base), t); \
        else fn<C, ##__VA_ARGS__, false>(lds, tl, rows<false>(base), t);     \
    } while (0)
#define RCS_CALL0(fn, base)                                     \
    do {                                                        \
        if (split) fn<C, true>(lds, tl, rows<true>(base), t);   \
        else fn<C, false>(lds, tl, rows<false>(base), t);       \
    } while (0)
    // parity of the red / blue columns in the thread's row(s) (both rows of a split wave have the same)
    RCS_MEM int par(const Tile &tl, int base) const { return png_of(tl.filters, base + ja); }

    RCS_MEM void fetch(const Tile &tl, int A, LoadRegs &v) const
    {
        if (split) st_fetch<true>(tl, rows<true>(A), t, v);
        else st_fetch<false>(tl, rows<false>(A), t, v);
    }
    RCS_MEM void commit(rcs_lf lds, const Tile &tl, int A, const LoadRegs &v) const
    {
        if (split) st_commit<C, true>(lds, tl, rows<true>(A), t, v);
        else st_commit<C, false>(lds, tl, rows<false>(A), t, v);
    }
    RCS_MEM void i1(rcs_lf lds, const Tile &tl, int A) const
    {
        if (par(tl, A - L_S2 - R)) RCS_CALL(st_s2, A - L_S2 - R, 1); else RCS_CALL(st_s2, A - L_S2 - R, 0);
        RCS_CALL0(st_s40, A - L_S40 - R);
        RCS_CALL0(st_s1, A - L_S1 - R);
    }
    RCS_MEM void i2(rcs_lf lds, const Tile &tl, int A) const
    {
        if (par(tl, A - L_S3 - R)) RCS_CALL(st_s3, A - L_S3 - R, 1); else RCS_CALL(st_s3, A - L_S3 - R, 0);
        if (par(tl, A - L_S41 - R)) RCS_CALL(st_s41, A - L_S41 - R, 1); else RCS_CALL(st_s41, A - L_S41 - R, 0);
    }
    RCS_MEM void i3(rcs_lf lds, const Tile &tl, int A) const
    {
        if (par(tl, A - L_S42 - R)) RCS_CALL(st_s42, A - L_S42 - R, 1); else RCS_CALL(st_s42, A - L_S42 - R, 0);
    }
    RCS_MEM void i4(rcs_lf lds, const Tile &tl, int A) const
    {
        if (par(tl, A - L_S43 - R)) RCS_CALL(st_s43, A - L_S43 - R, 0); else RCS_CALL(st_s43, A - L_S43 - R, 1);
    }
    // iterations of a tile: A = R, 2R, ... while the write-back still has rows to do
    static RCS_MEM bool more(const Tile &tl, int A) { return A - L_S43 - R < tl.rows - BORDER; }
};

} // namespace rcs
