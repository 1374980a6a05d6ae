// art_amd/csrc/rcd.hip -- RCD demosaic for gfx950, v1 "arena" kernel.
//
// Replaces RawImageSource::rcd_demosaic (reference: rtengine/rcd_demosaic.cc:51-347).  One
// workgroup per REFERENCE tile (194x194, stride 176, origin (0,0); the tile grid is part of
// the result because VH_Dir/PQ_Dir are only defined on [4,n-4) of each tile and their
// undefined positions read as 0 / as the aliased lpf values, rcd_demosaic.cc:101-103,199).
// Work planes live in a per-workgroup HBM arena, zeroed per tile (a reference thread's
// first tile).  Pure stencil chain: phases separated by workgroup barriers.
#include <hip/hip_runtime.h>
#include "devmath.h"
#include "kernels.h"

namespace artgpu {

namespace {
constexpr int ts = RCD_TS;
constexpr int w1 = ts, w2 = 2 * ts, w3 = 3 * ts, w4 = 4 * ts;
constexpr int NT = RCD_THREADS;
constexpr float eps = 1e-5f, epssq = 1e-10f, scale = 65536.f;

// (row, it) advance incrementally: one integer division per phase and thread instead of one per item.
#define FOR_ITEMS(R0, R1, N)                                                                                                     \
    for (int _n = (N), _tot = ((R1) > (R0) ? ((R1) - (R0)) : 0) * _n, _t = tid, _d = _n > 0 ? _n : 1, _q = NT / _d, _r = NT - _q * _d, \
             row = (R0) + tid / _d, it = tid - (row - (R0)) * _d;                                                                \
         _t < _tot; _t += NT, row += _q, it += _r, row += (it >= _d), it -= (it >= _d) ? _d : 0)                                  \
        for (int _once = 1; _once; _once = 0)

__device__ __forceinline__ float hpf(const float *c, int i, int s1)
{
    // (c[-3s] - c[-s] - c[+s] + c[+3s]) - 3*(c[-2s] + c[+2s]) + 6*c[0], squared
    return sqr((c[i - 3 * s1] - c[i - s1] - c[i + s1] + c[i + 3 * s1]) - 3.f * (c[i - 2 * s1] + c[i + 2 * s1]) + 6.f * c[i]);
}
} // namespace

__global__ void __launch_bounds__(RCD_THREADS)
rcd_tiles_kernel(RcdArgs a)
{
    const int tid = threadIdx.x;
    float *const A = a.arena + (size_t)blockIdx.x * RCD_ARENA_FLOATS;
    constexpr int full = ts * ts, half = full / 2;
    float *const cfa = A;
    float *const rgb0 = A + full, *const rgb1 = A + 2 * full, *const rgb2 = A + 3 * full;
    float *const VH_Dir = A + 4 * full;
    float *const PQ_Dir = A + 5 * full, *const lpf = PQ_Dir;
    float *const P_CDiff_Hpf = PQ_Dir + half, *const Q_CDiff_Hpf = P_CDiff_Hpf + half;
    const unsigned filters = a.filters;
    const int W = a.W, H = a.H;
    const int tileSizeN = ts - 2 * RCD_BORDER;
#define FCT(r, c) fc(filters, (unsigned)((r) & 1), (unsigned)((c) & 1))

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int tr = tile / a.numTw, tc = tile - tr * a.numTw;
        const int rowStart = tr * tileSizeN, rowEnd = min(rowStart + ts, H);
        const int colStart = tc * tileSizeN, colEnd = min(colStart + ts, W);
        if (rowStart + RCD_BORDER == rowEnd - RCD_BORDER || colStart + RCD_BORDER == colEnd - RCD_BORDER) continue;
        const int tileRows = min(rowEnd - rowStart, ts), tilecols = min(colEnd - colStart, ts);

        {
            float4 *A4 = reinterpret_cast<float4 *>(A);
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int i = tid; i < RCD_ARENA_FLOATS / 4; i += NT) A4[i] = z;
        }
        __syncthreads();

        // tile load: cfa and the two native colour planes of each row (L126-131)
        FOR_ITEMS(0, tileRows, tilecols) {
            const int indx = row * ts + it;
            const float v = lim01(a.raw[(size_t)(rowStart + row) * a.raw_stride + colStart + it] / scale);
            const unsigned c0 = FCT(rowStart + row, colStart), c1 = FCT(rowStart + row, colStart + 1);
            cfa[indx] = v;
            if (c0 == 0 || c1 == 0) rgb0[indx] = v;
            if (c0 == 1 || c1 == 1) rgb1[indx] = v;
            if (c0 == 2 || c1 == 2) rgb2[indx] = v;
        }
        __syncthreads();

        // Step 1: VH_Dir on [4,rows-4) x [4,cols-4) (L135-166)
        FOR_ITEMS(4, tileRows - 4, tilecols - 8 > 0 ? tilecols - 8 : 0) {
            const int indx = row * ts + 4 + it;
            const float V_Stat = std_max(epssq, hpf(cfa, indx - w1, w1) + hpf(cfa, indx, w1) + hpf(cfa, indx + w1, w1));
            const float H_Stat = std_max(epssq, hpf(cfa, indx - 1, 1) + hpf(cfa, indx, 1) + hpf(cfa, indx + 1, 1));
            VH_Dir[indx] = V_Stat / (V_Stat + H_Stat);
        }
        // Step 2: low-pass at non-green sites (L169-175)
        FOR_ITEMS(2, tileRows - 2, ngroups(2, tilecols - 2, 2)) {
            const int col = 2 + (FCT(row, 0) & 1) + 2 * it;
            if (col < tilecols - 2) {
                const int indx = row * ts + col;
                lpf[indx / 2] = cfa[indx] +
                                0.5f * (cfa[indx - w1] + cfa[indx + w1] + cfa[indx - 1] + cfa[indx + 1]) +
                                0.25f * (cfa[indx - w1 - 1] + cfa[indx - w1 + 1] + cfa[indx + w1 - 1] + cfa[indx + w1 + 1]);
            }
        }
        // Step 4.0: P/Q diagonal high-pass (L213-218) -- independent of steps 1-3
        FOR_ITEMS(3, tileRows - 3, ngroups(3, tilecols - 3, 2)) {
            const int indx = row * ts + 3 + 2 * it;
            P_CDiff_Hpf[indx / 2] = hpf(cfa, indx, w1 + 1);
            Q_CDiff_Hpf[indx / 2] = hpf(cfa, indx, w1 - 1);
        }
        __syncthreads();

        // Step 3: green at red/blue sites (L178-206)
        FOR_ITEMS(4, tileRows - 4, ngroups(4, tilecols - 4, 2)) {
            const int col = 4 + (FCT(row, 0) & 1) + 2 * it;
            if (col < tilecols - 4) {
                const int indx = row * ts + col, lp = indx / 2;
                const float cfai = cfa[indx];
                const float cN1 = cfa[indx - w1], cN2 = cfa[indx - w2], cN3 = cfa[indx - w3], cN4 = cfa[indx - w4];
                const float cS1 = cfa[indx + w1], cS2 = cfa[indx + w2], cS3 = cfa[indx + w3], cS4 = cfa[indx + w4];
                const float cW1 = cfa[indx - 1], cW2 = cfa[indx - 2], cW3 = cfa[indx - 3], cW4 = cfa[indx - 4];
                const float cE1 = cfa[indx + 1], cE2 = cfa[indx + 2], cE3 = cfa[indx + 3], cE4 = cfa[indx + 4];
                const float N_Grad = eps + (fabsf(cN1 - cS1) + fabsf(cfai - cN2)) + (fabsf(cN1 - cN3) + fabsf(cN2 - cN4));
                const float S_Grad = eps + (fabsf(cN1 - cS1) + fabsf(cfai - cS2)) + (fabsf(cS1 - cS3) + fabsf(cS2 - cS4));
                const float W_Grad = eps + (fabsf(cW1 - cE1) + fabsf(cfai - cW2)) + (fabsf(cW1 - cW3) + fabsf(cW2 - cW4));
                const float E_Grad = eps + (fabsf(cW1 - cE1) + fabsf(cfai - cE2)) + (fabsf(cE1 - cE3) + fabsf(cE2 - cE4));
                const float lpfi = lpf[lp];
                const float N_Est = cN1 * (lpfi + lpfi) / (eps + lpfi + lpf[lp - w1]);
                const float S_Est = cS1 * (lpfi + lpfi) / (eps + lpfi + lpf[lp + w1]);
                const float W_Est = cW1 * (lpfi + lpfi) / (eps + lpfi + lpf[lp - 1]);
                const float E_Est = cE1 * (lpfi + lpfi) / (eps + lpfi + lpf[lp + 1]);
                const float V_Est = (S_Grad * N_Est + N_Grad * S_Est) / (N_Grad + S_Grad);
                const float H_Est = (W_Grad * E_Est + E_Grad * W_Est) / (E_Grad + W_Grad);
                const float VH_C = VH_Dir[indx];
                const float VH_N = 0.25f * ((VH_Dir[indx - w1 - 1] + VH_Dir[indx - w1 + 1]) + (VH_Dir[indx + w1 - 1] + VH_Dir[indx + w1 + 1]));
                const float VH_Disc = fabsf(0.5f - VH_C) < fabsf(0.5f - VH_N) ? VH_N : VH_C;
                rgb1[indx] = intp(VH_Disc, H_Est, V_Est);
            }
        }
        __syncthreads();

        // Step 4.1: PQ_Dir (aliases lpf, last read in step 3) (L221-227)
        FOR_ITEMS(4, tileRows - 4, ngroups(4, tilecols - 4, 2)) {
            const int col = 4 + (FCT(row, 0) & 1) + 2 * it;
            if (col < tilecols - 4) {
                const int indx = row * ts + col, i2 = indx / 2, i3 = (indx - w1 - 1) / 2, i4 = (indx + w1 - 1) / 2;
                const float P_Stat = std_max(epssq, P_CDiff_Hpf[i3] + P_CDiff_Hpf[i2] + P_CDiff_Hpf[i4 + 1]);
                const float Q_Stat = std_max(epssq, Q_CDiff_Hpf[i3 + 1] + Q_CDiff_Hpf[i2] + Q_CDiff_Hpf[i4]);
                PQ_Dir[i2] = P_Stat / (P_Stat + Q_Stat);
            }
        }
        __syncthreads();

        // Step 4.2: red/blue at blue/red sites (L230-258)
        FOR_ITEMS(4, tileRows - 4, ngroups(4, tilecols - 4, 2)) {
            const int col = 4 + (FCT(row, 0) & 1) + 2 * it;
            if (col < tilecols - 4) {
                const int indx = row * ts + col;
                const int c = 2 - (int)FCT(row, col);
                float *rc = c == 0 ? rgb0 : rgb2;
                const int pq = indx / 2, pq2 = (indx - w1 - 1) / 2, pq3 = (indx + w1 - 1) / 2;
                const float PQ_C = PQ_Dir[pq];
                const float PQ_N = 0.25f * (PQ_Dir[pq2] + PQ_Dir[pq2 + 1] + PQ_Dir[pq3] + PQ_Dir[pq3 + 1]);
                const float PQ_Disc = (fabsf(0.5f - PQ_C) < fabsf(0.5f - PQ_N)) ? PQ_N : PQ_C;
                const float rNW = rc[indx - w1 - 1], rNE = rc[indx - w1 + 1], rSW = rc[indx + w1 - 1], rSE = rc[indx + w1 + 1];
                const float g0 = rgb1[indx];
                const float NW_Grad = eps + fabsf(rNW - rSE) + fabsf(rNW - rc[indx - w3 - 3]) + fabsf(g0 - rgb1[indx - w2 - 2]);
                const float NE_Grad = eps + fabsf(rNE - rSW) + fabsf(rNE - rc[indx - w3 + 3]) + fabsf(g0 - rgb1[indx - w2 + 2]);
                const float SW_Grad = eps + fabsf(rNE - rSW) + fabsf(rSW - rc[indx + w3 - 3]) + fabsf(g0 - rgb1[indx + w2 - 2]);
                const float SE_Grad = eps + fabsf(rNW - rSE) + fabsf(rSE - rc[indx + w3 + 3]) + fabsf(g0 - rgb1[indx + w2 + 2]);
                const float NW_Est = rNW - rgb1[indx - w1 - 1];
                const float NE_Est = rNE - rgb1[indx - w1 + 1];
                const float SW_Est = rSW - rgb1[indx + w1 - 1];
                const float SE_Est = rSE - rgb1[indx + w1 + 1];
                const float P_Est = (NW_Grad * SE_Est + SE_Grad * NW_Est) / (NW_Grad + SE_Grad);
                const float Q_Est = (NE_Grad * SW_Est + SW_Grad * NE_Est) / (NE_Grad + SW_Grad);
                rc[indx] = g0 + intp(PQ_Disc, Q_Est, P_Est);
            }
        }
        __syncthreads();

        // Step 4.3: red/blue at green sites (L261-302)
        FOR_ITEMS(4, tileRows - 4, ngroups(4, tilecols - 4, 2)) {
            const int col = 4 + (FCT(row, 1) & 1) + 2 * it;
            if (col < tilecols - 4) {
                const int indx = row * ts + col;
                const float VH_C = VH_Dir[indx];
                const float VH_N = 0.25f * ((VH_Dir[indx - w1 - 1] + VH_Dir[indx - w1 + 1]) + (VH_Dir[indx + w1 - 1] + VH_Dir[indx + w1 + 1]));
                const float VH_Disc = (fabsf(0.5f - VH_C) < fabsf(0.5f - VH_N)) ? VH_N : VH_C;
                const float g0 = rgb1[indx];
                const float N1 = eps + fabsf(g0 - rgb1[indx - w2]);
                const float S1 = eps + fabsf(g0 - rgb1[indx + w2]);
                const float W1 = eps + fabsf(g0 - rgb1[indx - 2]);
                const float E1 = eps + fabsf(g0 - rgb1[indx + 2]);
                const float gN = rgb1[indx - w1], gS = rgb1[indx + w1], gW = rgb1[indx - 1], gE = rgb1[indx + 1];
#pragma unroll
                for (int c = 0; c <= 2; c += 2) {
                    float *rc = c == 0 ? rgb0 : rgb2;
                    const float rN = rc[indx - w1], rS = rc[indx + w1], rW = rc[indx - 1], rE = rc[indx + 1];
                    const float SNabs = fabsf(rN - rS);
                    const float EWabs = fabsf(rW - rE);
                    const float N_Grad = N1 + SNabs + fabsf(rN - rc[indx - w3]);
                    const float S_Grad = S1 + SNabs + fabsf(rS - rc[indx + w3]);
                    const float W_Grad = W1 + EWabs + fabsf(rW - rc[indx - 3]);
                    const float E_Grad = E1 + EWabs + fabsf(rE - rc[indx + 3]);
                    const float N_Est = rN - gN, S_Est = rS - gS, W_Est = rW - gW, E_Est = rE - gE;
                    const float V_Est = (N_Grad * S_Est + S_Grad * N_Est) / (N_Grad + S_Grad);
                    const float H_Est = (E_Grad * W_Est + W_Grad * E_Est) / (E_Grad + W_Grad);
                    rc[indx] = g0 + intp(VH_Disc, H_Est, V_Est);
                }
            }
        }
        __syncthreads();

        // write-back of the tile interior (L304-316)
        {
            const int r0 = RCD_BORDER, r1 = (rowEnd - RCD_BORDER) - rowStart;
            const int c0 = RCD_BORDER, c1 = (colEnd - RCD_BORDER) - colStart;
            FOR_ITEMS(r0, r1, c1 > c0 ? c1 - c0 : 0) {
                const int col = c0 + it, idx = row * ts + col;
                const size_t o = (size_t)(rowStart + row) * a.out_stride + colStart + col;
                a.red[o] = std_max(0.f, rgb0[idx] * scale);
                a.green[o] = std_max(0.f, rgb1[idx] * scale);
                a.blue[o] = std_max(0.f, rgb2[idx] * scale);
            }
        }
        __syncthreads();
    }
#undef FCT
}

hipError_t launch_rcd(const RcdArgs &a, int grid, hipStream_t stream)
{
    hipLaunchKernelGGL(rcd_tiles_kernel, dim3(grid), dim3(RCD_THREADS), 0, stream, a);
    return hipGetLastError();
}

} // namespace artgpu
