// art_amd/csrc/xtrans.hip -- X-Trans (Markesteijn) demosaic for gfx950.
//
// Replaces RawImageSource::xtrans_interpolate(passes, useCieLab) (reference: rtengine/xtrans_demosaic.cc:181-969),
// cielab (L41-116, x86-64 path) and xtransborder_interpolate (L122-173).  One workgroup per REFERENCE tile (114x114,
// origin (3,3), stride 98 -- the tile grid decides where each direction buffer is defined, so it is part of the
// result), and the workgroup has its CU to itself (1024 threads, 157 KB of LDS):
//   * ONE direction buffer (3 planes x 114 x 114 floats) is in LDS at a time.  Within a pass every phase reads and writes
//     one buffer only and buffer k + 4 starts as buffer k after pass 0, so buffer k is filled from the CFA, taken through
//     pass 0, stored as rgb[k], taken through the other passes, stored as rgb[k + 4], converted to the perceptual space
//     in place and differentiated there; rgb[k] comes back once for its own conversion + derivative.
//   * the per-workgroup HBM arena holds what has to outlive that: rgb[ndir] (planar: [dir][channel][114][114]) |
//     lab[3][106][106] | drv[ndir][104][104] in the reference's layout (L301-308).  greenminmax lives where the reference
//     keeps it (in the lab region); of the lab planes only the last direction's are stored, because the uint8
//     homogeneity maps alias them and the 5x5 sums read map bytes no one wrote.
//   * the homogeneity maps are bytes in LDS (a copy of those arena bytes with the counts on top); their 5x5 sums, the
//     per-pixel maximum and the average of the chosen directions are one pass per pixel over them.
// Every step reads only values produced by EARLIER steps (the in-place green recalculation of pass >= 1 reads green
// and interpolated R/B at green sites only), so each step is a parallel loop over the tile's sites and steps are
// separated by workgroup barriers (LDS-only ones inside the buffer loop).
#include <hip/hip_runtime.h>
#include <float.h>
#include <type_traits>
#include "devmath.h"
#include "kernels.h"

namespace artgpu {

namespace {
constexpr int TS = XTRANS_TS, TSH = TS / 2, NT = XTRANS_THREADS;
constexpr int LW = TS - 8;     // lab plane pitch
constexpr int DW = TS - 10;    // drv plane pitch
constexpr int PL = TS * TS;    // one colour plane of a direction buffer (the rgb buffers are kept PLANAR here:
                               // [dir][channel][row][col] -- every value is written before it is read, so only the
                               // lab/drv/homo regions need the reference's exact byte layout)

// the small tables of the kernel arguments, copied to LDS: indexed per lane from the argument segment each look-up is a memory round trip
struct XtTables { int xtrans[36], allhex0[3][3][8], allhex1[3][3][8], right_shift[3], cls[2][2][9], ncls[2]; };
typedef const __attribute__((address_space(3))) XtTables *xt_tab;
struct Geo {
    xt_tab a;
    __device__ int fcol(int row, int col) const { return a->xtrans[(row % 6) * 6 + col % 6]; }
    __device__ int isgreen(int row, int col) const { return a->xtrans[(row % 3) * 6 + col % 3] & 1; }
};

typedef __attribute__((address_space(3))) float *xt_lf;
typedef float xt_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) xt_f4 *xt_lf4;
typedef __attribute__((address_space(3))) unsigned char *xt_lb;
// workgroup barrier that orders LDS traffic only: a __syncthreads() would also wait for the stores to the arena
__device__ __forceinline__ void xt_lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ float limf(float v, float lo, float hi) { return std_max(lo, std_min(v, hi)); }

template <typename H>
__device__ __forceinline__ void hex_minmax(const float *pix, H hex, float &mn, float &mx)
{
    float minval = FLT_MAX, maxval = 0.f;
#pragma unroll
    for (int c = 0; c < 6; c++) {
        const float val = pix[hex[c]];
        minval = minval < val ? minval : val;
        maxval = maxval > val ? maxval : val;
    }
    mn = minval; mx = maxval;
}
__device__ __forceinline__ float cbrt_lut(const float *__restrict__ lut, int i) { return lut[i < 0 ? 0 : (i > 0x14000 - 1 ? 0x14000 - 1 : i)]; }
} // namespace

// LDS of a workgroup (it has the CU to itself): the three planes of one direction buffer, 156 KB; the homogeneity phases reuse it for the eight
// homogeneity maps (bytes, 104 KB) and strips of the eight derivative planes (XT_DR rows + two halo rows).
constexpr int XT_LDS_FLOATS = 3 * TS * TS;
constexpr int XT_DR = (XT_LDS_FLOATS - 8 * TS * TS / 4) / (8 * (TS - 10)) - 2;      // 13 derivative rows per strip beside the eight byte maps
#define FOR_T(N) for (int t = tid, _n = (N); t < _n; t += NT)
// the same walk with t / P and t % P kept up to date by additions (a division by a constant is a quarter-rate multiply-high plus a multiply back)
#define FOR_T_RC(N, P, r, c) for (int t = tid, _n = (N), r = tid / (P), c = tid - (tid / (P)) * (P); t < _n; t += NT, r += NT / (P) + (c >= (P) - NT % (P)), c += c >= (P) - NT % (P) ? NT % (P) - (P) : NT % (P))
#ifdef XT_PROFILE
#define XT_MARK(k) do { const long long _n2 = wall_clock64(); xt_acc[k] += _n2 - xt_last; xt_last = _n2; } while (0)
#else
#define XT_MARK(k) do { } while (0)
#endif

#ifndef XTRANS_MIN_WAVES
#define XTRANS_MIN_WAVES 4     // 128 VGPRs, one 1024-thread workgroup per CU: with the loads of a pixel batched the phases want registers, not waves (46.1 -> 38.8 ms)
#endif
__global__ void __launch_bounds__(XTRANS_THREADS, XTRANS_MIN_WAVES) xtrans_tiles_kernel(XtransArgs a)
{
    extern __shared__ float xt_lds[];
    const int tid = threadIdx.x;
    __shared__ XtTables s_tab;
    if (tid < 36) s_tab.xtrans[tid] = a.xtrans[tid];
    if (tid < 72) { (&s_tab.allhex0[0][0][0])[tid] = (&a.allhex0[0][0][0])[tid]; (&s_tab.allhex1[0][0][0])[tid] = (&a.allhex1[0][0][0])[tid]; }
    if (tid < 3) s_tab.right_shift[tid] = a.right_shift[tid];
    // Dense site lists for the phases that work on one site class only (4 of 9 pixels each: with a lane per pixel more than half of
    // every wave sat out).  The green layout has period 3 (`isgreen`), so a class is a list of (row, column) residues per 3 x 3 cell
    // and the tile is 38 x 38 cells.  cls[0]: the non-green sites, cls[1]: the greens off the solitary green's row and column (2x2 blocks).
    if (tid == 0) {
        int n0 = 0, n1 = 0;
        for (int rr = 0; rr < 3; rr++)
            for (int cc = 0; cc < 3; cc++) {
                if (!(a.xtrans[rr * 6 + cc] & 1)) { s_tab.cls[0][0][n0] = rr; s_tab.cls[0][1][n0] = cc; n0++; }
                if ((rr - a.sgrow % 3 + 3) % 3 != 0 && (cc - a.sgcol % 3 + 3) % 3 != 0) { s_tab.cls[1][0][n1] = rr; s_tab.cls[1][1][n1] = cc; n1++; }
            }
        s_tab.ncls[0] = n0; s_tab.ncls[1] = n1;
    }
    __syncthreads();
    constexpr int NCELL = TS / 3;
    static_assert(NCELL * 3 == TS, "the tile is a whole number of 3 x 3 cells");
    // site `t` of class `cls` in a tile whose origin is (top, left): tile-local row and column
#define XT_SITE(KC, t, top, left, r, c)                                                               \
    const int _sn = T->ncls[KC];                                                                       \
    int _cell;                                                                                         \
    if (_sn == 4) _cell = (t) >> 2; else _cell = (t) / _sn;   /* uniform: four sites per cell for every real sensor */ \
    const int _k = (t) - _cell * _sn, _bi = _cell / NCELL;                                             \
    const int r = 3 * _bi + (T->cls[KC][0][_k] - (top) % 3 + 3) % 3, c = 3 * (_cell - _bi * NCELL) + (T->cls[KC][1][_k] - (left) % 3 + 3) % 3
    const xt_tab T = (xt_tab)&s_tab;
    const Geo G{T};
    // With four sites of a class per cell (every real sensor) the class member a thread works on never changes -- site t = tid + i * NT is member
    // tid & 3 of cell (tid >> 2) + 256 i -- so everything XT_SITE and the phases derive from (row % 3, col % 3) is a per-thread constant and the
    // cell advances by (6 rows, 28 columns) of cells per iteration: no division, no table look-up per site.  FOR_SITES4 walks the cells;
    // SiteK holds what depends on the tile origin.
    struct SiteK { int dr, dc; unsigned fpack; };      // tile-local residues of the thread's class member; fcol for (cell row, cell column) parities, 2 bits each
    auto site_setup = [&](int KC, int top_, int left_) {
        SiteK q;
        const int rr = T->cls[KC][0][tid & 3], cc = T->cls[KC][1][tid & 3];     // row % 3, col % 3 of the member
        q.dr = (rr - top_ % 3 + 3) % 3; q.dc = (cc - left_ % 3 + 3) % 3;
        const int mpar = ((top_ + q.dr - rr) / 3) & 1, npar = ((left_ + q.dc - cc) / 3) & 1;      // row = rr + 3 (cell row + m0): row % 6 = rr + 3 (parity)
        q.fpack = 0;
#pragma unroll
        for (int ab = 0; ab < 4; ++ab)
            q.fpack |= (unsigned)T->xtrans[(rr + 3 * (((ab >> 1) + mpar) & 1)) * 6 + cc + 3 * (((ab & 1) + npar) & 1)] << (2 * ab);
        return q;
    };
    constexpr int SITE_DR = (NT / 4) / NCELL, SITE_DC = (NT / 4) % NCELL;      // the walk's step in cells: NT / 4 cells = (6 rows, 28 columns)
    static_assert(NT % 4 == 0 && SITE_DC > 0, "four sites per cell; the cell step is not a whole number of cell rows");
#define FOR_SITES4(Q, r, c, f)                                                                                              \
    for (int _bi = (tid >> 2) / NCELL, _cj = (tid >> 2) - _bi * NCELL; _bi < NCELL; _bi += SITE_DR + (_cj >= NCELL - SITE_DC), _cj += _cj >= NCELL - SITE_DC ? SITE_DC - NCELL : SITE_DC) \
        for (int r = 3 * _bi + (Q).dr, c = 3 * _cj + (Q).dc, f = (int)(((Q).fpack >> (2 * (((_bi & 1) << 1) | (_cj & 1)))) & 3u), _once = 1; _once; _once = 0)
    const int ndir = a.ndir, passes = a.passes;
    float *const buffer = a.arena + (size_t)blockIdx.x * a.arena_floats;
    float *const labbase = buffer + (size_t)TS * TS * (ndir * 3);
    float *const drvbase = buffer + (size_t)TS * TS * (ndir * 3 + 3);
    unsigned char *const homo = reinterpret_cast<unsigned char *>(labbase);       // [ndir][TS][TS]
    float *const gmm = labbase;                                                   // [TS][TSH][2]
    const int width = a.W, height = a.H;
    const size_t rs = a.raw_stride;
    const int sgrow = a.sgrow, sgcol = a.sgcol;
#define RGB(d, r, c) (buffer + (size_t)(d) * 3 * PL + (r) * TS + (c))
#define LAB(k, i, j) labbase[((k) * LW + (i)) * LW + (j)]
#define DRV(d, i, j) drvbase[((d) * DW + (i)) * DW + (j)]

#ifdef XT_PROFILE
    long long xt_acc[16] = {0}, xt_last = wall_clock64();
#endif
    // tiles from a shared counter (cleared per launch): the first one is the workgroup's index, the others are handed out in order --
    // thread 0 asks for the next one while this one is being worked on (a fixed share per workgroup left the last ones idle for a tile's
    // length at the end, and edge tiles are cheaper than full ones)
    __shared__ int s_next;
    for (int tile = blockIdx.x; tile < a.ntiles;) {
        if (tid == 0) s_next = a.counter ? (int)gridDim.x + atomicAdd(a.counter, 1) : tile + (int)gridDim.x;
        const int tyi = tile / a.ntx, txi = tile - tyi * a.ntx;
        const int top = 3 + tyi * (TS - 16), left = 3 + txi * (TS - 16);
        int mrow = min(top + TS, height - 3), mcol = min(left + TS, width - 3);

        // ---- clear the lab planes (greenminmax / homo live there)
        FOR_T(3 * LW * LW) labbase[t] = 0.f;
        __syncthreads(); XT_MARK(0);

        // ---- green min/max (L320-408): one item per (row, group of the row's non-green run)
        FOR_T(TS * 40) {
            const int r = t / 40, g = t - r * 40, row = top + r;
            if (row >= mrow) continue;
            int leftstart = left;
            for (; leftstart < mcol; leftstart++)
                if (!G.isgreen(row, leftstart)) break;
            const float *rawrow = a.raw + (size_t)row * rs;
            float mn, mx;
            if (T->right_shift[row % 3]) {
                const int col = leftstart + 3 * g;
                if (col < mcol) {
                    hex_minmax(rawrow + col, T->allhex0[row % 3][col % 3], mn, mx);
                    float *s = gmm + ((size_t)r * TSH + ((col - left) >> 1)) * 2;
                    s[0] = mn; s[1] = mx;
                }
            } else {
                const int single = (G.fcol(row, leftstart + 1) & 1);    // coloffset == 2: the run starts with a lone pixel
                if (single && g == 0) {
                    hex_minmax(rawrow + leftstart, T->allhex0[row % 3][leftstart % 3], mn, mx);
                    float *s = gmm + ((size_t)r * TSH + ((leftstart - left) >> 1)) * 2;
                    s[0] = mn; s[1] = mx;
                } else {
                    const int col = leftstart + (single ? 2 : 0) + 3 * (g - single);
                    if (col < mcol) {
                        hex_minmax(rawrow + col, T->allhex0[row % 3][col % 3], mn, mx);   // the pair shares the first pixel's hexagon
                        float *s = gmm + ((size_t)r * TSH + ((col - left) >> 1)) * 2;
                        s[0] = mn; s[1] = mx;
                        if (col < mcol - 1) {
                            float *s2 = gmm + ((size_t)r * TSH + ((col + 1 - left) >> 1)) * 2;
                            s2[0] = mn; s2[1] = mx;
                        }
                    }
                }
            }
        }
        __syncthreads(); XT_MARK(1);

        // ---- The interpolation passes, the perceptual space and the derivative (L410-741), ONE DIRECTION BUFFER AT A TIME IN LDS.
        // Within a pass every phase reads and writes one buffer only, and buffer k + 4 is buffer k after pass 0 (the memcpy of L479-481),
        // so buffer k's three planes (3 x 114 x 114 floats = 156 KB: the workgroup has the CU to itself) are filled from the CFA, taken
        // through pass 0, stored as rgb[k], taken through the remaining passes and stored as rgb[k + 4]; cielab then overwrites the planes
        // in place and the derivative reads it there.  rgb[k] comes back once for its own cielab + derivative.  Offsets keep the arena's
        // meaning: plane stride PL, row stride TS.
        const int mrl = mrow - top, mcl = mcol - left;   // tile-local bounds (L654-655)
        const int nlab = mrl - 8;                        // lab rows [0, nlab)
        const xt_lf L = (xt_lf)xt_lds;
        const int row0s = (top - sgrow + 4) / 3 * 3 + sgrow, col0s = (left - sgcol + 4) / 3 * 3 + sgcol;
        const bool fast4 = T->ncls[0] == 4 && T->ncls[1] == 4;

        // cielab (L41-116) in place over rows 4 .. 4 + nlab, columns 4 .. 4 + LW of the buffer in LDS + the derivative along direction d (L657-741)
        auto lab_and_derivative = [&](int d) {
            if (a.use_cielab) {
                // four pixels per thread and iteration: their twelve table look-ups are in flight together
                constexpr int U = 4;
                const int n = nlab * LW;
                int wi = tid / LW, wj = tid - wi * LW;              // (row, column) of the walk's next pixel: + NT = + (9 rows, 70 columns)
                for (int t0 = tid; t0 < n; t0 += U * NT) {
                    int pp[U], jj[U], ii[U], ix[U][3];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const bool inside = t0 + u * NT < n;
                        const int i = inside ? wi : nlab - 1, j = inside ? wj : LW - 1;
                        wi += NT / LW + (wj >= LW - NT % LW); wj += wj >= LW - NT % LW ? NT % LW - LW : NT % LW;
                        const int p = (4 + i) * TS + 4 + j;
                        pp[u] = p; jj[u] = j; ii[u] = i;
                        const float p0 = L[p], p1 = L[PL + p], p2 = L[2 * PL + p];
                        // 4-lane groups while j < labWidth - 3 ...
                        const float x0 = p0 * a.xyz_cam[0] + p1 * a.xyz_cam[1] + p2 * a.xyz_cam[2];
                        const float x1 = p0 * a.xyz_cam[3] + p1 * a.xyz_cam[4] + p2 * a.xyz_cam[5];
                        const float x2 = p0 * a.xyz_cam[6] + p1 * a.xyz_cam[7] + p2 * a.xyz_cam[8];
                        ix[u][0] = __float2int_rn(x0); ix[u][1] = __float2int_rn(x1); ix[u][2] = __float2int_rn(x2);
                        // ... the scalar tail (the row's last two columns) rounds by adding 0.5 and truncating: only the waves that hold such a pixel
                        const bool vec = j < ((LW - 3 + 3) / 4) * 4;
                        if (__builtin_amdgcn_ballot_w64(!vec) != 0) {
                            float y0 = 0.5f, y1 = 0.5f, y2 = 0.5f;
                            y0 += a.xyz_cam[0] * p0; y1 += a.xyz_cam[3] * p0; y2 += a.xyz_cam[6] * p0;
                            y0 += a.xyz_cam[1] * p1; y1 += a.xyz_cam[4] * p1; y2 += a.xyz_cam[7] * p1;
                            y0 += a.xyz_cam[2] * p2; y1 += a.xyz_cam[5] * p2; y2 += a.xyz_cam[8] * p2;
                            if (!vec) { ix[u][0] = (int)y0; ix[u][1] = (int)y1; ix[u][2] = (int)y2; }
                        }
                    }
                    float cv[U][3];
#pragma unroll
                    for (int u = 0; u < U; u++)
#pragma unroll
                        for (int k = 0; k < 3; k++) cv[u][k] = cbrt_lut(a.cbrt_lut, ix[u][k]);
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        if (t0 + u * NT >= n) break;
                        const float Lv = 116.f * cv[u][1] - 16.f;
                        const float A = 500.f * (cv[u][0] - cv[u][1]), Bv = 200.f * (cv[u][1] - cv[u][2]);
                        L[pp[u]] = Lv; L[PL + pp[u]] = A; L[2 * PL + pp[u]] = Bv;
                        // the last direction's planes are stored as well: the homogeneity maps alias them (L301-308), and the 5x5 sums
                        // read map bytes no one wrote = bytes of those floats
                        if (d == ndir - 1) { const int i = ii[u]; LAB(0, i, jj[u]) = Lv; LAB(1, i, jj[u]) = A; LAB(2, i, jj[u]) = Bv; }
                    }
                }
            } else {
                FOR_T_RC(nlab * LW, LW, i, j) {
                    if (j >= mcl - 8) continue;
                    const int p = (4 + i) * TS + 4 + j;
                    const float p0 = L[p], p1 = L[PL + p], p2 = L[2 * PL + p];
                    const float y = 0.2627f * p0 + 0.6780f * p1 + 0.0593f * p2;
                    const float A = (p2 - y) * 0.56433f, Bv = (p0 - y) * 0.67815f;
                    L[p] = y; L[PL + p] = A; L[2 * PL + p] = Bv;
                    if (d == ndir - 1) { LAB(0, i, j) = y; LAB(1, i, j) = A; LAB(2, i, j) = Bv; }
                }
            }
            xt_lds_barrier();
            const int dd = d & 3;
            const int f = dd == 0 ? 1 : (dd == 1 ? TS : (dd == 2 ? TS + 1 : TS - 1));
            FOR_T_RC((mrl - 10) * TS, TS, rr, c) {
                const int r = 5 + rr;
                if (c < 5 || c >= mcl - 5) continue;
                const xt_lf l = L + r * TS + c, aa = l + PL, b = l + 2 * PL;
                float v;
                if (a.use_cielab) {
                    const float g = 2 * l[0] - l[f] - l[-f];
                    v = sqr(g) + sqr((2 * aa[0] - aa[f] - aa[-f] + g * 2.1551724f)) + sqr((2 * b[0] - b[f] - b[-f] - g * 0.86206896f));
                } else {
                    v = sqr(2 * l[0] - l[f] - l[-f]) + sqr(2 * aa[0] - aa[f] - aa[-f]) + sqr(2 * b[0] - b[f] - b[-f]);
                }
                DRV(d, r - 5, c - 5) = v;
            }
            xt_lds_barrier();
        };
        // the buffer in LDS -> rgb[d] of the arena
        auto store_buffer = [&](int d) {
            xt_f4 *dst = reinterpret_cast<xt_f4 *>(buffer + (size_t)d * 3 * PL);
            FOR_T(3 * PL / 4) dst[t] = ((xt_lf4)L)[t];
        };

        for (int k = 0; k < 4; k++) {
            // ---- CFA samples, green interpolated along direction k at the non-green sites (L410-475).  Four pixels per thread and
            // iteration with their loads first (the sample, four neighbours along the hexagon entry buffer k takes, greenminmax): one pixel
            // at a time is a chain of global round trips per iteration.  Buffer k holds colour j = k ^ flip (gdir[j ^ flip] = color[j]):
            //   j = 0: 0.68 (p[h1] + p[h0]) - 0.18 (p[2 h1] + p[2 h0]);  j = 1: 0.87 p[h3] + 0.13 p[h2] + 0.36 (p[0] - p[-h2]);
            //   j >= 2, h = hex[2 + j]: 0.64 p[h] + 0.36 p[-2 h] + 0.13 (2 p[0] - p[3 h] - p[-3 h])
            for (int t0 = tid; t0 < TS * TS; t0 += 4 * NT) {
                float cen[4], nb[4][4], lo[4], hi[4];
                int fq[4], jq[4];
                bool inq[4], itq[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int t = min(t0 + u * NT, TS * TS - 1);
                    const int r = t / TS, c = t - r * TS, row = top + r, col = left + c;
                    const bool in = row < mrow && col < mcol;
                    const int rowc = in ? row : top, colc = in ? col : left;       // a pixel outside the frame reads the tile's first one (unused)
                    const int f = G.fcol(rowc, colc);
                    const bool it = in && !(f & 1);
                    const auto hex = T->allhex0[rowc % 3][colc % 3];
                    const int flip = T->right_shift[rowc % 3] ? 0 : 1;
                    const int j = k ^ flip;
                    int o0 = 0, o1 = 0, o2 = 0, o3 = 0;
                    if (it) {
                        if (j == 0) { o0 = hex[1]; o1 = hex[0]; o2 = 2 * hex[1]; o3 = 2 * hex[0]; }
                        else if (j == 1) { o0 = hex[3]; o1 = hex[2]; o2 = -hex[2]; }
                        else { const int h = hex[2 + j]; o0 = h; o1 = -2 * h; o2 = 3 * h; o3 = -3 * h; }
                    }
                    const float *pix = a.raw + (size_t)rowc * rs + colc;
                    cen[u] = pix[0]; nb[u][0] = pix[o0]; nb[u][1] = pix[o1]; nb[u][2] = pix[o2]; nb[u][3] = pix[o3];
                    const float *sm = gmm + ((size_t)r * TSH + (c >> 1)) * 2;
                    lo[u] = sm[0]; hi[u] = sm[1];
                    fq[u] = f; jq[u] = j; inq[u] = in; itq[u] = it;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int t = t0 + u * NT;
                    if (t >= TS * TS) break;
                    float base[3] = {0.f, 0.f, 0.f};
                    if (inq[u]) base[fq[u]] = cen[u];
                    float color;
                    if (jq[u] == 0) color = 0.6796875f * (nb[u][0] + nb[u][1]) - 0.1796875f * (nb[u][2] + nb[u][3]);
                    else if (jq[u] == 1) color = 0.87109375f * nb[u][0] + nb[u][1] * 0.12890625f + 0.359375f * (cen[u] - nb[u][2]);
                    else color = 0.640625f * nb[u][0] + 0.359375f * nb[u][1] + 0.12890625f * (2.f * cen[u] - nb[u][2] - nb[u][3]);
                    L[t] = base[0]; L[PL + t] = itq[u] ? limf(color, lo[u], hi[u]) : base[1]; L[2 * PL + t] = base[2];
                }
            }
            xt_lds_barrier();

            for (int pass = 0; pass < passes; pass++) {
                // (set up per pass: held across the whole tile the site constants push the final phases into scratch)
                const SiteK q0 = site_setup(0, top, left), q1 = site_setup(1, top, left);
                // recalculate green from interpolated values of closer pixels (L483-524): buffer k is the target of hexagon entry
                // (k ^ flip) + 2, of none where k == flip
                if (pass) {
                    if (fast4) {
                        const int rr0 = T->cls[0][0][tid & 3], cc0 = T->cls[0][1][tid & 3];
                        const int flip = T->right_shift[rr0] ? 0 : 1, e = k ^ flip;
                        const int hx = T->allhex1[rr0][cc0][e + 2];
                        if (e != 0) {
                            FOR_SITES4(q0, r, c, f) {
                                if (r < 2 || c < 2 || r >= mrl - 2 || c >= mcl - 2) continue;
                                const float *s = gmm + ((size_t)r * TSH + (c >> 1)) * 2;
                                const xt_lf rix = L + r * TS + c;
                                const float val = 0.33333333f * (rix[-2 * hx + PL] + 2 * (rix[hx + PL] - rix[hx + f * PL]) - rix[-2 * hx + f * PL]) + rix[f * PL];
                                rix[PL] = limf(val, s[0], s[1]);
                            }
                        }
                    } else {
                    FOR_T(NCELL * NCELL * T->ncls[0]) {
                        XT_SITE(0, t, top, left, r, c);
                        const int row = top + r, col = left + c;
                        if (r < 2 || c < 2 || row >= mrow - 2 || col >= mcol - 2) continue;
                        const int flip = T->right_shift[row % 3] ? 0 : 1;
                        const int e = k ^ flip;
                        if (e == 0) continue;
                        const int f = G.fcol(row, col);
                        const int hx = T->allhex1[row % 3][col % 3][e + 2];
                        const float *s = gmm + ((size_t)r * TSH + (c >> 1)) * 2;
                        const xt_lf rix = L + r * TS + c;
                        const float val = 0.33333333f * (rix[-2 * hx + PL] + 2 * (rix[hx + PL] - rix[hx + f * PL]) - rix[-2 * hx + f * PL]) + rix[f * PL];
                        rix[PL] = limf(val, s[0], s[1]);
                    }
                    }
                    xt_lds_barrier();
                }
                // red and blue for solitary green pixels (L527-561): buffers 0 and 1 take the row / column pair as it is, buffers 2
                // and 3 the better of the two
                FOR_T(40 * 40) {
                    const int i3 = t / 40, j3 = t - i3 * 40;
                    const int row = row0s + 3 * i3, col = col0s + 3 * j3;
                    if (row >= mrow - 2 || col >= mcol - 2) continue;
                    const int h0 = G.fcol(row, col0s + 1) ^ ((j3 & 1) ? 2 : 0);
                    const xt_lf rix = L + (row - top) * TS + (col - left);
                    const float gc = rix[PL];
                    float c0 = 0.f, c2 = 0.f, pc0 = 0.f, pc2 = 0.f, pdiff = 0.f;
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        if (k < 2 && q) break;
                        const int d = k < 2 ? k : 2 * (k - 1) + q;                 // 0 | 1 | 2, 3 | 4, 5
                        const int i = (d & 1) ? TS : 1, hd = h0 ^ ((d & 1) ? 2 : 0);
                        float ck[2], diff = 0.f;
#pragma unroll
                        for (int kk = 0; kk < 2; kk++) {
                            const int o = i << kk, hk = hd ^ (kk ? 2 : 0);
                            const float gp = rix[o + PL], gm = rix[-o + PL], hp = rix[o + hk * PL], hm = rix[-o + hk * PL];
                            const float g = gc + gc - gp - gm;
                            ck[kk] = g + hp + hm;
                            diff += sqr(gp - gm - hp + hm) + sqr(g);
                        }
                        c0 = hd == 0 ? ck[0] : ck[1];
                        c2 = hd == 0 ? ck[1] : ck[0];
                        if (q && pdiff < diff) { c0 = pc0; c2 = pc2; }
                        pc0 = c0; pc2 = c2; pdiff = diff;
                    }
                    rix[0] = 0.5f * c0;
                    rix[2 * PL] = 0.5f * c2;
                }
                xt_lds_barrier();
                // red for blue pixels and vice versa (L564-606): only buffer dc ever compares the two candidate axes
                if (fast4) {
                    const int cd = ((T->cls[0][0][tid & 3] - sgrow % 3 + 3) % 3) ? TS : 1;         // (row - sgrow) % 3 != 0
                    const int hd = 3 * (cd ^ TS ^ 1);
                    const bool both = k == (cd == 1 ? 1 : 0);
                    FOR_SITES4(q0, r, c, fc) {
                        if (r < 3 || c < 3 || r >= mrl - 3 || c >= mcl - 3) continue;
                        const int f = 2 - fc;
                        const xt_lf rix = L + r * TS + c;
                        const float g0 = rix[PL];
                        int i = cd;
                        if (both && !((fabsf(g0 - rix[cd + PL]) + fabsf(g0 - rix[-cd + PL])) < 2.f * (fabsf(g0 - rix[hd + PL]) + fabsf(g0 - rix[-hd + PL])))) i = hd;
                        rix[f * PL] = g0 + 0.5f * (rix[i + f * PL] + rix[-i + f * PL] - rix[i + PL] - rix[-i + PL]);
                    }
                } else {
                FOR_T(NCELL * NCELL * T->ncls[0]) {
                    XT_SITE(0, t, top, left, r, c);
                    const int row = top + r, col = left + c;
                    if (r < 3 || c < 3 || row >= mrow - 3 || col >= mcol - 3) continue;
                    const int cd = ((row - sgrow) % 3) ? TS : 1;
                    const int hd = 3 * (cd ^ TS ^ 1);
                    const int f = 2 - G.fcol(row, col);
                    const xt_lf rix = L + r * TS + c;
                    const int dc = cd == 1 ? 1 : 0;
                    const float g0 = rix[PL];
                    int i = cd;
                    if (k == dc && !((fabsf(g0 - rix[cd + PL]) + fabsf(g0 - rix[-cd + PL])) < 2.f * (fabsf(g0 - rix[hd + PL]) + fabsf(g0 - rix[-hd + PL])))) i = hd;
                    rix[f * PL] = g0 + 0.5f * (rix[i + f * PL] + rix[-i + f * PL] - rix[i + PL] - rix[-i + PL]);
                }
                }
                xt_lds_barrier();
                // red and blue for 2x2 blocks of green (L609-650): the reference steps d by two over the hexagon table while it steps by
                // one buffer, so with four directions only buffers 0 and 1 are filled
                if (2 * k < ndir) {
                    if (fast4) {
                        const int rr1 = T->cls[1][0][tid & 3], cc1 = T->cls[1][1][tid & 3];
                        const int h0 = T->allhex1[rr1][cc1][2 * k], h1 = T->allhex1[rr1][cc1][2 * k + 1];
                        const bool third = (h0 + h1) != 0;
                        FOR_SITES4(q1, r, c, fu) {
                            (void)fu;
                            if (r < 2 || c < 2 || r >= mrl - 2 || c >= mcl - 2) continue;
                            const xt_lf rix = L + r * TS + c;
                            const float gc = rix[PL], g0 = rix[h0 + PL], g1 = rix[h1 + PL], r0 = rix[h0], r1 = rix[h1], b0 = rix[h0 + 2 * PL], b1 = rix[h1 + 2 * PL];
                            if (third) {
                                const float g = 3 * gc - 2 * g0 - g1;
                                rix[0] = (g + 2 * r0 + r1) * 0.33333333f; rix[2 * PL] = (g + 2 * b0 + b1) * 0.33333333f;
                            } else {
                                const float g = 2 * gc - g0 - g1;
                                rix[0] = (g + r0 + r1) * 0.5f; rix[2 * PL] = (g + b0 + b1) * 0.5f;
                            }
                        }
                    } else {
                    FOR_T(NCELL * NCELL * T->ncls[1]) {
                        XT_SITE(1, t, top, left, r, c);
                        const int row = top + r, col = left + c;
                        if (r < 2 || c < 2 || row >= mrow - 2 || col >= mcol - 2) continue;
                        const auto hex = T->allhex1[row % 3][col % 3];
                        const int h0 = hex[2 * k], h1 = hex[2 * k + 1];
                        const xt_lf rix = L + r * TS + c;
                        if (h0 + h1) {
                            const float g = 3 * rix[PL] - 2 * rix[h0 + PL] - rix[h1 + PL];
                            const float vr = (g + 2 * rix[h0] + rix[h1]) * 0.33333333f;
                            const float vb = (g + 2 * rix[h0 + 2 * PL] + rix[h1 + 2 * PL]) * 0.33333333f;
                            rix[0] = vr; rix[2 * PL] = vb;
                        } else {
                            const float g = 2 * rix[PL] - rix[h0 + PL] - rix[h1 + PL];
                            const float vr = (g + rix[h0] + rix[h1]) * 0.5f;
                            const float vb = (g + rix[h0 + 2 * PL] + rix[h1 + 2 * PL]) * 0.5f;
                            rix[0] = vr; rix[2 * PL] = vb;
                        }
                    }
                    }
                    xt_lds_barrier();
                }
                if (pass == 0) { store_buffer(k); xt_lds_barrier(); }
                else if (pass == passes - 1) { store_buffer(k + 4); xt_lds_barrier(); }
            }
            XT_MARK(2 + k);
            lab_and_derivative(passes > 1 ? k + 4 : k);
            if (passes > 1) {
                // rgb[k] once more, for its own cielab + derivative (written above by this workgroup: a full barrier)
                __syncthreads();
                const xt_f4 *src = reinterpret_cast<const xt_f4 *>(buffer + (size_t)k * 3 * PL);
                FOR_T(3 * PL / 4) ((xt_lf4)L)[t] = src[t];
                xt_lds_barrier();
                lab_and_derivative(k);
            }
            XT_MARK(6 + k);
        }
        __syncthreads();

        // ---- homogeneity maps (L744-811), their 5x5 sums (L823-866), the per-pixel maximum (L870-906) and the average of the most
        // homogeneous directions (L910-949).  The byte maps live in LDS: first a copy of what the arena holds where the reference keeps
        // them (the bytes of the last direction's lab planes / greenminmax / the cleared rest: the 5x5 sums at the frame's edges read map
        // bytes no one wrote), then the counts on top of it; the sums, the maximum and the average are one pass per pixel over that.
        // The derivative rows of all directions are staged beside the maps strip by strip (every value is read 9 x for the counts and
        // once for the threshold).
        const xt_lb s_b = (xt_lb)xt_lds;                                  // [ndir][TS][TS]
        const xt_lf sdrv = L + 8 * TS * TS / 4;                           // [ndir][XT_DR + 2][DW]
        {
            const unsigned *src = reinterpret_cast<const unsigned *>(homo);                 // TS * TS is a multiple of 4
            FOR_T(ndir * (TS * TS / 4)) ((__attribute__((address_space(3))) unsigned *)xt_lds)[t] = src[t];
        }
        // (a strip's values are fetched into registers while the strip before it is counted).  Thread = (direction, column): tid >> 7 and
        // tid & 127 -- 104 of 128 lanes hold a column --, the strip's rows are a loop with constant strides: no index arithmetic per element
        // (element = tid + u NT needed (direction, row, column) from divisions by the strip's run-time row count).
        static_assert(DW <= 128 && NT >= 8 * 128, "one thread per (direction, derivative column)");
        const int sd = tid >> 7, sj = tid & 127;
        const bool sact = sd < ndir && sj < DW;
        float pre[XT_DR + 2];
        auto fetch_strip = [&](int ra) {
            const int rb = min(ra + XT_DR, mrl - 6), nrows = rb - ra + 2;
            const float *col = &DRV(sact ? sd : 0, ra - 6, sact ? sj : 0);
#pragma unroll
            for (int ii = 0; ii < XT_DR + 2; ii++) pre[ii] = col[min(ii, nrows - 1) * DW];
        };
        if (6 < mrl - 6) fetch_strip(6);
        for (int ra = 6; ra < mrl - 6; ra += XT_DR) {
            const int rb = min(ra + XT_DR, mrl - 6), nrows = rb - ra + 2;       // derivative rows ra - 6 .. rb - 5
            if (sact) {
#pragma unroll
                for (int ii = 0; ii < XT_DR + 2; ii++)
                    if (ii < nrows) sdrv[(sd * (XT_DR + 2) + ii) * DW + sj] = pre[ii];
            }
            if (ra + XT_DR < mrl - 6) fetch_strip(ra + XT_DR);
            xt_lds_barrier();
            // Two horizontally adjacent pixels per thread: their 3 x 3 windows share two of three columns (12 LDS reads per direction for
            // the pair instead of 18), and with the direction loop unrolled (ndir is 4 or 8) the reads of all directions are issued
            // together instead of one LDS round trip per direction.
            constexpr int NPAIR = (TS - 12 + 1) / 2;                  // columns 6 .. TS - 7 in pairs
            auto count_pairs = [&](auto nd_) {
                constexpr int ND = decltype(nd_)::value;
                FOR_T((rb - ra) * NPAIR) {
                    const int rr = t / NPAIR, pc = t - rr * NPAIR, r = ra + rr, c = 6 + 2 * pc;
                    if (c >= mcl - 6) continue;
                    const bool two = c + 1 < mcl - 6;
                    const xt_lf base = sdrv + (rr + 1) * DW + c - 5;                 // derivative (row, column) of the pair's first pixel
                    float cen[ND][2];
#pragma unroll
                    for (int d = 0; d < ND; d++) { cen[d][0] = base[d * (XT_DR + 2) * DW]; cen[d][1] = base[d * (XT_DR + 2) * DW + 1]; }
                    float tr0 = cen[0][0] < cen[1][0] ? cen[0][0] : cen[1][0], tr1 = cen[0][1] < cen[1][1] ? cen[0][1] : cen[1][1];
#pragma unroll
                    for (int d = 2; d < ND; d++) { tr0 = (cen[d][0] < tr0 ? cen[d][0] : tr0); tr1 = (cen[d][1] < tr1 ? cen[d][1] : tr1); }
                    tr0 *= 8; tr1 *= 8;
#pragma unroll
                    for (int d0 = 0; d0 < ND; d0 += 4) {          // four directions' windows (48 values) in flight at a time
                        float x[4][3][4];
#pragma unroll
                        for (int d = 0; d < 4; d++)
#pragma unroll
                            for (int v = 0; v < 3; v++)
#pragma unroll
                                for (int h = 0; h < 4; h++) x[d][v][h] = base[((d0 + d) * (XT_DR + 2) + v - 1) * DW + h - 1];
#pragma unroll
                        for (int d = 0; d < 4; d++) {
                            int cnt0 = 0, cnt1 = 0;
#pragma unroll
                            for (int v = 0; v < 3; v++)
#pragma unroll
                                for (int h = 0; h < 3; h++) { cnt0 += (x[d][v][h] <= tr0 ? 1 : 0); cnt1 += (x[d][v][h + 1] <= tr1 ? 1 : 0); }
                            s_b[((d0 + d) * TS + r) * TS + c] = (unsigned char)cnt0;
                            if (two) s_b[((d0 + d) * TS + r) * TS + c + 1] = (unsigned char)cnt1;
                        }
                    }
                }
            };
            if (ndir == 8) count_pairs(std::integral_constant<int, 8>{});
            else count_pairs(std::integral_constant<int, 4>{});
            xt_lds_barrier();
        }
        XT_MARK(10);

        int mr2 = mrl, mc2 = mcl;
        if (height - top < TS + 4) mr2 = height - top + 2;
        if (width - left < TS + 4) mc2 = width - left + 2;
        const int startrow = min(top, 8), startcol = min(left, 8);
        FOR_T_RC(TS * TS, TS, r, c) {
            if (r < startrow || c < startcol || r >= mr2 - 8 || c >= mc2 - 8) continue;
            // the reference's 16-wide loop adds with unsigned saturation (_mm_adds_epu8, L835-843); its running-sum tail,
            // which only the last row reaches, truncates to uint8 (L846-864).  Sums above 255 arise where never-written
            // homogeneity bytes (= lab bytes) are read at the right / bottom edge.
            const int endcol = r < mr2 - 9 ? mc2 - 8 : mc2 - 23;
            const int ncov = endcol > startcol ? ((endcol - startcol + 15) / 16) * 16 : 0;
            const bool saturate = c < startcol + ncov;
            unsigned char hm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            // The 5 x 5 byte sums, a row of five as the two aligned words that hold it: the first four bytes through v_alignbyte + v_sad_u8,
            // the fifth extracted from the upper word (200 single-byte LDS reads per pixel were what this phase took its time for).  A direction's
            // map is a whole number of words, so the alignment of a row is the same in all eight; it alternates with the row (TS = 2 mod 4).
            unsigned rowbase[5], rowsh[5];
#pragma unroll
            for (int v = 0; v < 5; v++) {
                const unsigned A = (unsigned)((r + v - 2) * TS + c - 2);
                rowbase[v] = A >> 2; rowsh[v] = A & 3u;
            }
            static_assert((TS * TS) % 4 == 0, "a homogeneity map is a whole number of words");
            const __attribute__((address_space(3))) unsigned *const s_w = (const __attribute__((address_space(3))) unsigned *)xt_lds;
#pragma unroll
            for (int d = 0; d < 8; d++) {
                if (d >= ndir) break;
                unsigned sum = 0;
#pragma unroll
                for (int v = 0; v < 5; v++) {
                    const __attribute__((address_space(3))) unsigned *const w = s_w + d * (TS * TS / 4) + rowbase[v];
                    const unsigned lo = w[0], hi = w[1];
                    sum = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(hi, lo, rowsh[v]), 0u, sum);
                    sum += (hi >> (8u * rowsh[v])) & 0xffu;
                }
                hm[d] = saturate ? (unsigned char)(sum > 255u ? 255u : sum) : (unsigned char)sum;
            }
            unsigned char maxval = hm[0];
#pragma unroll
            for (int d = 1; d < 8; d++)
                if (d < ndir) maxval = maxval < hm[d] ? hm[d] : maxval;
            maxval -= maxval >> 3;
            if (ndir > 4) {
#pragma unroll
                for (int d = 4; d < 8; d++) {
                    if (hm[d - 4] < hm[d]) hm[d - 4] = 0;
                    else if (hm[d - 4] > hm[d]) hm[d] = 0;
                }
            }
            float avg[4] = {0.f, 0.f, 0.f, 0.f};
            // the chosen directions' colours: every load is issued (a direction that is not chosen re-reads direction 0's address and adds
            // +0, which changes no bit of a sum that starts at +0) instead of one guarded round trip per direction
            float pv[8][3];
#pragma unroll
            for (int d = 0; d < 8; d++) {
                const bool on = d < ndir && hm[d] >= maxval;
                const float *p = RGB(on ? d : 0, r, c);
                pv[d][0] = p[0]; pv[d][1] = p[PL]; pv[d][2] = p[2 * PL];
            }
#pragma unroll
            for (int d = 0; d < 8; d++) {
                const bool on = d < ndir && hm[d] >= maxval;
                avg[0] += on ? pv[d][0] : 0.f; avg[1] += on ? pv[d][1] : 0.f; avg[2] += on ? pv[d][2] : 0.f;
                avg[3] += on ? 1.f : 0.f;
            }
            const size_t o = (size_t)(r + top) * a.out_stride + c + left;
            a.red[o] = std_max(0.f, avg[0] / avg[3]);
            a.green[o] = std_max(0.f, avg[1] / avg[3]);
            a.blue[o] = std_max(0.f, avg[2] / avg[3]);
        }
        __syncthreads(); XT_MARK(13);
        tile = s_next;
        __syncthreads();
    }
#ifdef XT_PROFILE
    if (blockIdx.x == 7 && tid == 0) {
        printf("XTPROF");
        for (int k = 0; k < 14; ++k) printf(" %lld", xt_acc[k]);
        printf("\n");
    }
#endif
}

// xtransborder_interpolate (L122-173): one lane per frame pixel inside the border strips
__global__ void __launch_bounds__(256) xtrans_border_kernel(XtransArgs a)
{
    const int width = a.W, height = a.H, border = a.border;
    const long long n = (long long)width * height;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(t / width), col = (int)(t - (long long)row * width);
        if (row >= border && row < height - border && col >= border && col < width - border) continue;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, w0 = 0.f, w1 = 0.f, w2 = 0.f;   // sum[0..2], sum[3..5]
        for (int y = max(0, row - 1); y <= min(row + 1, height - 1); y++)
            for (int x = max(0, col - 1); x <= min(col + 1, width - 1); x++) {
                const int f = a.xtrans[(y % 6) * 6 + x % 6];
                const float wgt = (y == row && x == col) ? 0.f : ((y == row || x == col) ? 0.5f : 0.25f);
                const float v = a.raw[(size_t)y * a.raw_stride + x] * wgt;
                if (f == 0) { s0 += v; w0 += wgt; }
                else if (f == 1) { s1 += v; w1 += wgt; }
                else { s2 += v; w2 += wgt; }
            }
        const float here = a.raw[(size_t)row * a.raw_stride + col];
        const size_t o = (size_t)row * a.out_stride + col;
        const int fc_ = a.xtrans[(row % 6) * 6 + col % 6];
        if (fc_ == 0) {
            a.red[o] = here; a.green[o] = s1 / w1; a.blue[o] = s2 / w2;
        } else if (fc_ == 1) {
            if (w0 == 0.f) { a.red[o] = here; a.green[o] = here; a.blue[o] = here; }
            else { a.red[o] = s0 / w0; a.green[o] = here; a.blue[o] = s2 / w2; }
        } else {
            a.red[o] = s0 / w0; a.green[o] = s1 / w1; a.blue[o] = here;
        }
    }
}

hipError_t launch_xtrans(const XtransArgs &a, int grid, hipStream_t s)
{
    if (hipError_t e = dyn_lds_once(reinterpret_cast<const void *>(&xtrans_tiles_kernel), XT_LDS_FLOATS * 4); e != hipSuccess) return e;
    hipLaunchKernelGGL(xtrans_tiles_kernel, dim3(grid), dim3(XTRANS_THREADS), XT_LDS_FLOATS * 4, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const long long n = (long long)a.W * a.H;
    long long g = (n + 255) / 256;
    XtransArgs b = a;
    b.border = a.passes > 1 ? 8 : 11;   // xtrans_demosaic.cc:968
    hipLaunchKernelGGL(xtrans_border_kernel, dim3((unsigned)(g < 16384 ? g : 16384)), dim3(256), 0, s, b);
    return hipGetLastError();
}

} // namespace artgpu
