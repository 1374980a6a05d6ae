// art_amd/csrc/wavelet.hip -- rtengine::wavelet_decomposition for subsampling == 1 on gfx950
// (reference: rtengine/cplx_wavelet_dec.h:97-270, cplx_wavelet_level.h:206-744).
//
//   level 0      decimated Daub4 (6 taps incl. two zero taps, offset 2): vertical then horizontal,
//                fused in one kernel: the vertically filtered rows (the reference's tmpLo/tmpHi)
//                live in LDS, the input is read ~1.2x, the four half-size subbands written once.
//   levels >= 1  undecimated Haar, skip = 2^(level-1): 4 taps per output, one lane per pixel.
//   synthesis    mirrors: Haar levels fused (H then V) through a ping-pong low-pass buffer,
//                level 0 fused with the horizontally synthesised rows in LDS, x4 and blend.
// All tap sums keep the reference's accumulation order (zero taps included) -> bit-exact.
// HBM-bound: analysis0 4 B/px in + 4 B/px out; Haar level 4 B/px(sub) in + 16 B out.
#include <hip/hip_runtime.h>
#include "devmath.h"
#include "kernels.h"

namespace artgpu {

namespace {
__constant__ float DAUB_LO[6] = {0.f, 0.f, 0.34150635f, 0.59150635f, 0.15849365f, -0.091506351f};
__constant__ float DAUB_HI[6] = {-0.091506351f, -0.15849365f, 0.59150635f, -0.34150635f, 0.f, 0.f};
// synthesis filters = reversed analysis filters (cplx_wavelet_dec.h:113-115)
__constant__ float SYN_LO[6] = {-0.091506351f, 0.15849365f, 0.59150635f, 0.34150635f, 0.f, 0.f};
__constant__ float SYN_HI[6] = {0.f, 0.f, -0.34150635f, 0.59150635f, -0.15849365f, -0.091506351f};
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

#ifndef A0_TW_
#define A0_TW_ 64
#endif
#ifndef A0_TH_
#ifdef A0_PLAIN
#define A0_TH_ 16
#else
#define A0_TH_ 32
#endif
#endif
constexpr int A0_TW = A0_TW_, A0_TH = A0_TH_;   // output tile of the level-0 analysis (64 x 32; -DA0_PLAIN: the six-loads-per-value form of rounds 1 - 4 with its 64 x 16)
constexpr int A0_LW = 2 * A0_TW + 6;           // tmp columns kept in LDS
#ifndef S0_TW_
#define S0_TW_ 128
#endif
#ifndef S0_TH_
#define S0_TH_ 32
#endif
constexpr int S0_TW = S0_TW_, S0_TH = S0_TH_;   // output tile of the level-0 synthesis
constexpr int S0_LH = S0_TH / 2 + 4;           // horizontally synthesised rows kept in LDS
} // namespace

// ---- level 0 analysis: src (w x h) -> lo, b1, b2, b3 (w2 x h2) ----
// Column stage first (the reference's tmpLo / tmpHi, here a tile in LDS), then the row stage from LDS.  A tmp value is six input rows of one column,
// and output rows one apart share four of their six: a thread walks A0_WR output rows down ONE tmp column with the 2 A0_WR + 4 input values in
// registers -- all loads issued first, 2.25 loads per tmp value instead of 6 --, 128 columns x A0_TH / A0_WR row groups per workgroup; the six halo
// columns of the tile take the plain six-load form.  Every sum keeps the reference's term order (zero taps included).
#ifdef A0_PLAIN
__global__ void __launch_bounds__(256) wavelet_analysis0_kernel(WaveArgs a)
{
    __shared__ float tLo[A0_TH][A0_LW], tHi[A0_TH][A0_LW];
    const int w = a.w, h = a.h, w2 = a.w2;
    const int c0 = blockIdx.x * A0_TW, r0 = blockIdx.y * A0_TH; // output coords
    const int icol0 = 2 * c0 - 3;                                // first tmp column held
    for (int t = threadIdx.x; t < A0_TH * A0_LW; t += 256) {
        const int rr = t / A0_LW, cc = t - rr * A0_LW;
        const int orow = r0 + rr;
        if (orow >= a.h2) continue;
        const int row = 2 * orow;
        const int k = clampi(icol0 + cc, 0, w - 1);
        float l = 0.f, hh = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float s = a.src[(size_t)clampi(row + (2 - j), 0, h - 1) * a.src_stride + k];
            l += DAUB_LO[j] * s;
            hh += DAUB_HI[j] * s;
        }
        tLo[rr][cc] = l;
        tHi[rr][cc] = hh;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < A0_TH * A0_TW; t += 256) {
        const int rr = t / A0_TW, cc = t - rr * A0_TW;
        const int orow = r0 + rr, ocol = c0 + cc;
        if (orow >= a.h2 || ocol >= w2) continue;
        const int i = 2 * ocol;
        float l0 = 0.f, h0 = 0.f, l1 = 0.f, h1 = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            // clamped column i+2-j, expressed in LDS coordinates (LDS column x holds clamp(icol0+x))
            const int x = clampi(i + (2 - j), 0, w - 1) - icol0;
            const float sl = tLo[rr][x], sh = tHi[rr][x];
            l0 += DAUB_LO[j] * sl; h0 += DAUB_HI[j] * sl;
            l1 += DAUB_LO[j] * sh; h1 += DAUB_HI[j] * sh;
        }
        const size_t o = (size_t)orow * w2 + ocol;
        a.lo[o] = l0; a.b1[o] = h0; a.b2[o] = l1; a.b3[o] = h1;
    }
}

#else
constexpr int A0_WR = 16;
static_assert(A0_TW == 64 && A0_TH * 128 == 256 * A0_WR, "the walkers' mapping: 128 tmp columns x A0_TH / A0_WR row groups = 256 threads");
__global__ void __launch_bounds__(256) wavelet_analysis0_kernel(WaveArgs a)
{
    __shared__ float tLo[A0_TH][A0_LW], tHi[A0_TH][A0_LW];
    const int w = a.w, h = a.h, w2 = a.w2;
    const int c0 = blockIdx.x * A0_TW, r0 = blockIdx.y * A0_TH; // output coords
    const int icol0 = 2 * c0 - 3;                                // first tmp column held
    {
        // tmp columns 0 .. 127: thread = (row group, column)
        const int g = threadIdx.x >> 7, cc = threadIdx.x & 127;
        const int orow0 = r0 + g * A0_WR;
        const int k = clampi(icol0 + cc, 0, w - 1);
        const int b0 = 2 * orow0 - 3;                            // lowest input row of the walk: output row q reads rows b0 + 2 q + 5 - j, j = 0 .. 5
        float v[2 * A0_WR + 4];
#pragma unroll
        for (int i = 0; i < 2 * A0_WR + 4; ++i) v[i] = a.src[(size_t)clampi(b0 + i, 0, h - 1) * a.src_stride + k];
        // tmp columns 128 .. 133 (A0_TH x 6 values): one each
        const int hr = threadIdx.x / 6, hc = 128 + (int)threadIdx.x - hr * 6;
        const bool halo = hr < A0_TH;
        const int hk = clampi(icol0 + hc, 0, w - 1), hrow = 2 * (r0 + (halo ? hr : 0));
        float hv[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) hv[j] = a.src[(size_t)clampi(hrow + (2 - j), 0, h - 1) * a.src_stride + hk];
#pragma unroll
        for (int q = 0; q < A0_WR; ++q) {
            float l = 0.f, hh = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float sv = v[2 * q + 5 - j];
                l += DAUB_LO[j] * sv;
                hh += DAUB_HI[j] * sv;
            }
            if (orow0 + q < a.h2) {
                tLo[g * A0_WR + q][cc] = l;
                tHi[g * A0_WR + q][cc] = hh;
            }
        }
        if (halo && r0 + hr < a.h2) {
            float l = 0.f, hh = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                l += DAUB_LO[j] * hv[j];
                hh += DAUB_HI[j] * hv[j];
            }
            tLo[hr][hc] = l;
            tHi[hr][hc] = hh;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < A0_TH * A0_TW; t += 256) {
        const int rr = t / A0_TW, cc = t - rr * A0_TW;
        const int orow = r0 + rr, ocol = c0 + cc;
        if (orow >= a.h2 || ocol >= w2) continue;
        const int i = 2 * ocol;
        float l0 = 0.f, h0 = 0.f, l1 = 0.f, h1 = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            // clamped column i+2-j, expressed in LDS coordinates (LDS column x holds clamp(icol0+x))
            const int x = clampi(i + (2 - j), 0, w - 1) - icol0;
            const float sl = tLo[rr][x], sh = tHi[rr][x];
            l0 += DAUB_LO[j] * sl; h0 += DAUB_HI[j] * sl;
            l1 += DAUB_LO[j] * sh; h1 += DAUB_HI[j] * sh;
        }
        const size_t o = (size_t)orow * w2 + ocol;
        a.lo[o] = l0; a.b1[o] = h0; a.b2[o] = l1; a.b3[o] = h1;
    }
}
#endif

// ---- levels >= 1 analysis: undecimated Haar (cplx_wavelet_level.h:206-238) ----
__global__ void __launch_bounds__(256) wavelet_haar_analysis_kernel(WaveArgs a)
{
    const int w = a.w2, h = a.h2, skip = a.skip;
    const long long n = (long long)w * h;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(t / w), i = (int)(t - (long long)row * w);
        const int rp = row < h - skip ? row + skip : row - skip;
        const int ip = i < w - skip ? i + skip : i - skip;
        const float s00 = a.src[(size_t)row * w + i], s01 = a.src[(size_t)row * w + ip];
        const float s10 = a.src[(size_t)rp * w + i], s11 = a.src[(size_t)rp * w + ip];
        const float tl0 = 0.25f * (s00 + s10), tl1 = 0.25f * (s01 + s11);
        const float th0 = 0.25f * (s00 - s10), th1 = 0.25f * (s01 - s11);
        a.lo[t] = tl0 + tl1; a.b1[t] = tl0 - tl1;
        a.b2[t] = th0 + th1; a.b3[t] = th0 - th1;
    }
}

// ---- levels >= 1 synthesis, fused H then V (cplx_wavelet_level.h:243-298) ----
__device__ __forceinline__ float haar_syn(float lo, float hi, float lop, float hip, bool first)
{
    return first ? (lo + hi) : 0.5f * (lo + hi + lop - hip);
}
__global__ void __launch_bounds__(256) wavelet_haar_synthesis_kernel(WaveArgs a)
{
    const int w = a.w2, h = a.h2, skip = a.skip;
    const long long n = (long long)w * h;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(t / w), j = (int)(t - (long long)row * w);
        const bool fj = j < skip, fr = row < skip;
        const int jp = fj ? j : j - skip, rp = fr ? row : row - skip;
        const size_t o00 = (size_t)row * w + j, o01 = (size_t)row * w + jp, o10 = (size_t)rp * w + j, o11 = (size_t)rp * w + jp;
        // tmpLo = synthH(src, b1), tmpHi = synthH(b2, b3) at (row, j) and (row-skip, j)
        const float tLo0 = haar_syn(a.src[o00], a.b1[o00], a.src[o01], a.b1[o01], fj);
        const float tHi0 = haar_syn(a.b2[o00], a.b3[o00], a.b2[o01], a.b3[o01], fj);
        float r;
        if (fr) {
            r = tLo0 + tHi0;
        } else {
            const float tLo1 = haar_syn(a.src[o10], a.b1[o10], a.src[o11], a.b1[o11], fj);
            const float tHi1 = haar_syn(a.b2[o10], a.b3[o10], a.b2[o11], a.b3[o11], fj);
            r = 0.5f * (tLo0 + tHi0 + tLo1 - tHi1);
        }
        a.lo[t] = r;
    }
}

// ---- level 0 synthesis (cplx_wavelet_level.h:449-584): H (w2 -> w) into LDS, V (h2 -> h), x4, blend ----
__global__ void __launch_bounds__(256) wavelet_synthesis0_kernel(WaveArgs a)
{
    __shared__ float tLo[S0_LH][S0_TW], tHi[S0_LH][S0_TW];
    const int w = a.w, h = a.h, w2 = a.w2, h2 = a.h2;
    const int c0 = blockIdx.x * S0_TW, r0 = blockIdx.y * S0_TH;
    constexpr int shift = 3; // taps - offset - 1
    const int srow0 = (r0 + shift) / 2 - 2; // first (unclamped) source row held in LDS
    // Horizontal pass.  Outputs i = 2q+1 and i = 2q+2 read the same three source columns (q, q+1, q+2: i_src = q+2 for both,
    // the odd one takes taps 0,2,4 and the even one taps 1,3,5), so one item computes the pair from one set of loads.
    // The tile starts at an even column: pair p covers tile columns 2p-1 and 2p, p = 0..S0_TW/2.
    constexpr int NP = S0_TW / 2 + 1;
    // All loads of a thread's items are issued before any arithmetic, from clamped (always valid) addresses: with the items as a loop whose
    // body starts with a validity test the loads of item n + 1 waited for item n's LDS stores -- five to six dependent memory round trips
    // per tile and thread (the kernel ran at 2.2 TB/s of its bytes).
    constexpr int NI = (S0_LH * NP + 255) / 256, NH = (NI + 1) / 2;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float vs[NH][3], v1[NH][3], v2[NH][3], v3[NH][3];
#pragma unroll
        for (int n = 0; n < NH; ++n) {
            const int t = min((int)threadIdx.x + (half * NH + n) * 256, S0_LH * NP - 1);
            const int rr = t / NP, p = t - rr * NP;
            const int i1 = min(c0 + 2 * p - 1, w - 1);
            const int k = clampi(srow0 + rr, 0, h2 - 1);
            const int i_src = (i1 + shift) / 2;
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const size_t arg = (size_t)k * w2 + clampi(i_src - l, 0, w2 - 1);
                vs[n][l] = a.src[arg]; v1[n][l] = a.b1[arg]; v2[n][l] = a.b2[arg]; v3[n][l] = a.b3[arg];
            }
        }
#pragma unroll
        for (int n = 0; n < NH; ++n) {
            const int t = (int)threadIdx.x + (half * NH + n) * 256;
            if (half * NH + n >= NI || t >= S0_LH * NP) continue;
            const int rr = t / NP, p = t - rr * NP;
            const int cc1 = 2 * p - 1, cc2 = 2 * p;                  // tile columns of the odd / even output
            const int i1 = c0 + cc1;                                 // odd (c0 is even)
            if (i1 >= w) continue;
            float lo1 = 0.f, hi1 = 0.f, lo2 = 0.f, hi2 = 0.f;
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const float s = vs[n][l], d1 = v1[n][l], d2 = v2[n][l], d3 = v3[n][l];
                lo1 += (SYN_LO[2 * l] * s + SYN_HI[2 * l] * d1);
                hi1 += (SYN_LO[2 * l] * d2 + SYN_HI[2 * l] * d3);
                lo2 += (SYN_LO[2 * l + 1] * s + SYN_HI[2 * l + 1] * d1);
                hi2 += (SYN_LO[2 * l + 1] * d2 + SYN_HI[2 * l + 1] * d3);
            }
            if (cc1 >= 0) { tLo[rr][cc1] = lo1; tHi[rr][cc1] = hi1; }
            if (cc2 < S0_TW && i1 + 1 < w) { tLo[rr][cc2] = lo2; tHi[rr][cc2] = hi2; }
        }
    }
    __syncthreads();
    const float srcFactor = 1.f - a.blend;
    // the destination values that take part in the blend are loaded up front: as `dst[o] = dst[o] * f + ...` in the loop every
    // iteration's load had to wait for the previous iteration's store (same array), sixteen dependent memory round trips per thread
    // With blend == 1 (every reconstruct() of RGB_denoise) the old value only contributes `old * 0`: nothing for a finite old value (the sum
    // below is never -0, so the sign of that zero cannot show), and a non-finite pixel has already sent the reference's MadRgb histogram index
    // out of range (FTblockDN.cc:587).  The load is skipped then -- a quarter of this kernel's traffic -- and the destination may be a plane
    // that was never written (the L channel is reconstructed into a second plane so that the first one stays as `Lin`).
    constexpr int NIT = S0_TH * S0_TW / 256;
    // The taps of output row i are filter entries begin, begin + 2, begin + 4 with begin = (i + shift) % 2.  A thread's rows all have one
    // parity (r0 and the 256 / S0_TW row step are even), so its three tap pairs are chosen once -- indexed per output they were six loads
    // from the constant table per output (a per-lane index is a vector load), half of this kernel's memory instructions.
    static_assert((256 / S0_TW) % 2 == 0 && S0_TH % 2 == 0, "row parity per thread");
    const bool odd_tap = ((r0 + (int)threadIdx.x / S0_TW + shift) & 1) != 0;
    float flo[3], fhi[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) { flo[l] = odd_tap ? SYN_LO[2 * l + 1] : SYN_LO[2 * l]; fhi[l] = odd_tap ? SYN_HI[2 * l + 1] : SYN_HI[2 * l]; }
    float dv[NIT];
    const bool use_old = a.blend != 1.f;
#pragma unroll
    for (int n = 0; n < NIT; ++n) {
        const int t = threadIdx.x + n * 256;
        const int rr = t / S0_TW, cc = t - rr * S0_TW;
        const int i = min(r0 + rr, h - 1), k = min(c0 + cc, w - 1);
        dv[n] = use_old ? a.dst[(size_t)i * a.dst_stride + k] : 0.f;
    }
#pragma unroll
    for (int n = 0; n < NIT; ++n) {
        const int t = threadIdx.x + n * 256;
        const int rr = t / S0_TW, cc = t - rr * S0_TW;
        const int i = r0 + rr, k = c0 + cc;
        if (i >= h || k >= w) continue;
        const int i_src = (i + shift) / 2;
        float tot = 0.f;
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            // LDS row rr holds source row clamp(srow0 + rr); the clamped row R sits at R - srow0 (0..18)
            const int lr = clampi(i_src - l, 0, h2 - 1) - srow0;
            tot += (flo[l] * tLo[lr][cc] + fhi[l] * tHi[lr][cc]);
        }
        const size_t o = (size_t)i * a.dst_stride + k;
        a.dst[o] = dv[n] * srcFactor + a.blend * 4.f * tot;
    }
}

hipError_t launch_wavelet_analysis0(const WaveArgs &a, hipStream_t s)
{
    dim3 grid((a.w2 + A0_TW - 1) / A0_TW, (a.h2 + A0_TH - 1) / A0_TH);
    hipLaunchKernelGGL(wavelet_analysis0_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}
static int flat_grid(long long n) { long long g = (n + 255) / 256; return (int)(g < 16384 ? g : 16384); }
hipError_t launch_wavelet_haar_analysis(const WaveArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(wavelet_haar_analysis_kernel, dim3(flat_grid((long long)a.w2 * a.h2)), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_wavelet_haar_synthesis(const WaveArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(wavelet_haar_synthesis_kernel, dim3(flat_grid((long long)a.w2 * a.h2)), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_wavelet_synthesis0(const WaveArgs &a, hipStream_t s)
{
    dim3 grid((a.w + S0_TW - 1) / S0_TW, (a.h + S0_TH - 1) / S0_TH);
    hipLaunchKernelGGL(wavelet_synthesis0_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

} // namespace artgpu
