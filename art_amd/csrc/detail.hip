// art_amd/csrc/detail.hip -- DCT detail recovery of RGB_denoise on gfx950
// (reference: rtengine/FTblockDN.cc:1479-1635 detail_recovery, L494-525 RGBtile_denoise,
//  L531-558 RGBoutput_tile_row; rtengine/boxblur.h:745-886 boxabsblur), luminanceDetailThreshold == 0.
//
// The reference hands the 64x64 block transforms to FFTW3 (REDFT10 / REDFT01 plans made with
// FFTW_MEASURE, L1604,1614,1924-1931): a third-party library whose round-off is not reproducible,
// so this stage is compared with a tolerance, everything else on the path is
// bit-exact.  Here one WAVE owns one block: each lane keeps a 64-sample line in registers and
// runs a straight-line fast DCT-II / DCT-III on it (dct64.h, Lee's recursion: ~770 operations per
// line instead of 4096, two per packed fp32 instruction), with one LDS transpose between the two dimensions of each transform.
// ~73 k blocks per 45 MP frame; VALU-bound.
// Block results go to a block buffer; a second kernel sums the up-to-9 overlapping blocks per
// pixel in the reference's serial order (vblk, then hblk) -- deterministic, no float atomics.
#include <hip/hip_runtime.h>
#include "devmath.h"
#include "devsleef.h"
#include "dct64.h"
#include "kernels.h"

namespace artgpu {

namespace {
constexpr int TS = 64, OFF = 25, BLKRAD = 1;
constexpr int NXCD = 8;                         // XCDs of an MI355X (workgroups are dispatched to them round robin)
__device__ __forceinline__ int reflect(int v, int n)
{
    // datarow / row padding of detail_recovery (L1545-1562)
    if (v < 0) return -v < n - 1 ? -v : n - 1;
    if (v >= n) return 2 * n - 2 - v > 0 ? 2 * n - 2 - v : 0;
    return v;
}
} // namespace

template <int RAD>
__global__ void __launch_bounds__(64) detail_blocks_kernel(DetailArgs a)
{
    // One 64 x 65 LDS buffer (+4 spare rows) per wave: the coefficients themselves stay in registers between the forward and
    // the inverse transform (lane = coefficient row), so nine blocks fit a CU instead of four.
    constexpr int XR = 4;                       // spare rows, >= blur radius + 1
    __shared__ float B[TS + XR][TS + 1];
    const int lane = threadIdx.x;
    // Blocks overlap 64 / 25 = 2.56 x per axis, so a pixel of Lin / L is wanted by up to nine blocks: 2.4 GB of loads for 0.36 GB of
    // planes on a 45 MP frame.  Workgroups are dealt to the eight XCDs round robin (workgroup b runs on XCD b % 8, each with an L2 of its
    // own), so in raster order every XCD's L2 saw every image row in flight -- the ~7 block rows the chip holds at once = 240 image rows x
    // the whole width = 16 MB against 4 MB of L2 -- and nearly every load missed (counters: 3.7 GB per launch).  Here XCD k owns the
    // k-th eighth of the block COLUMNS and walks it in raster order: its L2 sees 240 rows x an eighth of the width = 2 MB, and the
    // re-reads stay on the XCD.  (Which workgroup computes a block does not change the block: same bits.)
    const int wk = (a.numblox_W + NXCD - 1) / NXCD;
    const int xcd = blockIdx.x % NXCD, idx = blockIdx.x / NXCD;
    const int vblk = idx / wk, hblk = xcd * wk + (idx - vblk * wk);
    if (hblk >= a.numblox_W) return;            // (the ragged eighth; uniform over the workgroup)
    const int blk = vblk * a.numblox_W + hblk;
    const int top = (vblk - BLKRAD) * OFF, left = (hblk - BLKRAD) * OFF;
    constexpr int rad = RAD;                    // = a.blur_rad, 1..3 (XR - 1)
    float x[TS];

    // 1. load: lane = column j; x[i] = tilemask_in[i][j] * (Lin - L)(top+i, left+j), reflected
    {
        const int cc = reflect(left + lane, a.w);
#pragma unroll
        for (int i = 0; i < TS; ++i) {
            const int rr = reflect(top + i, a.h);
            const size_t o = (size_t)rr * a.w + cc;
            x[i] = a.tm_in[i * TS + lane] * (a.Lin[o] - a.L[o]);
        }
    }
    // 2. REDFT10 along i (= 2 * DCT-II) -> B[k][j]
    lee_fwd<TS>(x);
#pragma unroll
    for (int k = 0; k < TS; ++k) B[k][lane] = 2.f * x[k];
    __syncthreads();
    // 3. REDFT10 along j: lane = row k; the coefficients T[k][m] stay in x[m]
#pragma unroll
    for (int j = 0; j < TS; ++j) x[j] = B[lane][j];
    lee_fwd<TS>(x);
#pragma unroll
    for (int m = 0; m < TS; ++m) x[m] = 2.f * x[m];
    // 4. boxabsblur of the coefficients, radius `rad` (boxblur.h:745-886): along the row, straight from the registers into
    //    this lane's own row of B (which only this lane has read)
    {
        int len = rad + 1;
        float tempval = fabsf(x[0]);
#pragma unroll
        for (int q = 1; q <= rad; q++) tempval += fabsf(x[q]);
        tempval /= len;
        B[lane][0] = tempval;
#pragma unroll
        for (int col = 1; col <= rad; col++) {
            tempval = (tempval * len + fabsf(x[col + rad])) / (len + 1);
            B[lane][col] = tempval;
            len++;
        }
        const float rlen = 1.f / (float)len;
#pragma unroll
        for (int col = rad + 1; col < TS - rad; col++) {
            tempval = tempval + (fabsf(x[col + rad]) - fabsf(x[col - rad - 1])) * rlen;
            B[lane][col] = tempval;
        }
#pragma unroll
        for (int col = TS - rad; col < TS; col++) {
            tempval = (tempval * len - fabsf(x[col - rad - 1])) / (len - 1);
            B[lane][col] = tempval;
            len--;
        }
    }
    __syncthreads();
    // ... then down the columns (lane = column m) and the shrink factor 1 - exp(-blur^2 / factor).  The factor of row `row`
    // is written where N[row - rad - 1] was (its last use is this step); the first rad + 1 rows use the spare rows.
    // Round 5, two changes that only pay together (scripts/r5_ab8.sh: 0.91 ms per 45 MP frame before, the same with either one alone, 0.73 with both):
    //  * the steady rows go in groups of eight inside a ROLLED loop -- the sixteen LDS reads of a group first (they do not depend on the running sum),
    //    then the eight dependent additions, then the eight factors, whose writes land on rows the group has already read; one row per
    //    iteration waited an LDS round trip per row with two waves per SIMD to hide it.  (All 64 rows as straight-line code ran as fast alone
    //    but slowed the frame: 50 KB of code beside the chroma reconstructions' kernels on the side stream.)
    //  * -blur^2 / factor as a multiplication by the factor's reciprocal, formed once per lane for the launch's two factors: the correctly
    //    rounded division was 17 of a row's ~50 issue slots.  Two roundings instead of one (the reciprocal, then the product): the quotient can
    //    differ from the correctly rounded one by up to ~1.5 units in the last place, the shrink factor by ~2e-7 of itself -- like the exponential
    //    below, far inside what the stage's tolerance is there for (FFTW's own round-off).  With a detail MASK the factor differs per
    //    pixel: a per-pixel reciprocal is a division itself, so that branch keeps the reference's `/ factor` (no speed to gain, no bits to lose).
    {
        float lenf = (float)(rad + 1);
        float tv = B[0][lane];
#pragma unroll
        for (int i = 1; i <= rad; i++) tv = tv + B[i][lane];
        tv = tv / lenf;
        const bool colin = (left + lane) >= 0 && (left + lane) < a.w;
        const float inv_hi = 1.f / a.detail_hi, inv_lo = 1.f / a.detail_lo;
        // detail_factor is indexed by block position (FTblockDN.cc:1571-1596): hi inside the image
        auto emit = [&](int row, float tvr) {
            const bool rowin = (top + row) >= 0 && (top + row) < a.h;
            float factor = (rowin && colin) ? a.detail_hi : a.detail_lo;
            const float rfac = (rowin && colin) ? inv_hi : inv_lo;
            const bool masked = a.mask && rowin && colin;
            if (masked) {     // compute_detail(params_Ldetail * mask) (FTblockDN.cc:1481-1486,1583)
                const float d = a.params_Ldetail * a.mask[(size_t)(top + row) * a.w + left + lane];
                const float t = static_cast<float>((100. - d) * (100. - d) + 50. * (100. - d)) * TS * 0.5f;
                factor = t * t;
            }
            (void)factor; (void)rfac;
            const int slot = row > rad ? row - rad - 1 : TS + row;
            // (hardware exp2: the 4096 exponentials of a block were a quarter of the kernel's instructions as sleef's xexpf; this stage
            // is tolerance-checked against an exact DCT anyway -- FFTW's round-off is not reproducible -- and the factor changes by 2^-22)
#ifdef DETAIL_EXACT_DIV
            B[slot][lane] = 1.0f - __expf(-sqr(tvr) / factor);
#else
            if (a.mask) B[slot][lane] = 1.0f - __expf(masked ? -sqr(tvr) / factor : -sqr(tvr) * rfac);      // (uniform: only launches with a mask carry the division)
            else B[slot][lane] = 1.0f - __expf(-sqr(tvr) * rfac);
#endif
        };
        emit(0, tv);
#pragma unroll
        for (int row = 1; row <= rad; ++row) {
            const float lenp1 = lenf + 1.f;
            tv = (tv * lenf + B[row + rad][lane]) / lenp1;
            lenf = lenp1;
            emit(row, tv);
        }
        const float rlen = 1.f / lenf;
        constexpr int S0 = rad + 1, S1 = TS - rad, G = 8;          // steady rows [S0, S1)
        constexpr int SG = S0 + (S1 - S0) / G * G;                 // ... of which [S0, SG) in whole groups
#ifndef DETAIL_ROW_AT_A_TIME      // (defined: the steady rows one per iteration, the form before round 5 -- timing comparisons only)
#pragma unroll 1
        for (int r0 = S0; r0 < SG; r0 += G) {
            float hi[G], lo[G], tvs[G];
#pragma unroll
            for (int k = 0; k < G; ++k) { hi[k] = B[r0 + k + rad][lane]; lo[k] = B[r0 + k - rad - 1][lane]; }
#pragma unroll
            for (int k = 0; k < G; ++k) { tv = tv + (hi[k] - lo[k]) * rlen; tvs[k] = tv; }
#pragma unroll
            for (int k = 0; k < G; ++k) emit(r0 + k, tvs[k]);
        }
#pragma unroll 1
        for (int row = SG; row < S1; ++row) {
#else
#pragma unroll 1
        for (int row = S0; row < S1; ++row) {
#endif
            tv = tv + (B[row + rad][lane] - B[row - rad - 1][lane]) * rlen;
            emit(row, tv);
        }
#pragma unroll 1
        for (int row = S1; row < TS; ++row) {
            const float lenm1 = lenf - 1.f;
            tv = (tv * lenf - B[row - rad - 1][lane]) / lenm1;
            lenf = lenm1;
            emit(row, tv);
        }
    }
    __syncthreads();
    // 5. shrink (lane = row k again), then REDFT01 along m (= DCT-III of (X0, 2 X1, 2 X2, ...))
    {
        const int slot = lane > rad ? lane - rad - 1 : TS + lane;
#pragma unroll
        for (int m = 0; m < TS; ++m) x[m] = x[m] * B[slot][m];
    }
    __syncthreads();
#pragma unroll
    for (int m = 1; m < TS; ++m) x[m] = 2.f * x[m];
    lee_inv<TS>(x);
#pragma unroll
    for (int j = 0; j < TS; ++j) B[lane][j] = x[j];
    __syncthreads();
    // 6. REDFT01 along k: lane = column j; the result row i, column j goes straight to the block buffer (coalesced rows)
    x[0] = B[0][lane];
#pragma unroll
    for (int k = 1; k < TS; ++k) x[k] = 2.f * B[k][lane];
    lee_inv<TS>(x);
    float *out = a.blocks + (size_t)blk * TS * TS;
#pragma unroll
    for (int i = 0; i < TS; ++i) out[i * TS + lane] = x[i];
}

// Sum the overlapping blocks per pixel in the reference's serial order and add the detail to L:
//   Ldetail += tilemask_out * block * DCTnorm ; totwt += tilemask_in * tilemask_out ; L += Ldetail / totwt
__global__ void __launch_bounds__(1024) detail_gather_kernel(DetailArgs a)
{
    // the two 64 x 64 tile masks in LDS: every term of a pixel's sum looks both up at the pixel's position inside the block -- as global
    // loads they were two of the three memory instructions per term (18 of 27 per pixel)
    __shared__ float s_in[TS * TS], s_out[TS * TS];
    for (int k = threadIdx.x; k < TS * TS; k += 1024) { s_in[k] = a.tm_in[k]; s_out[k] = a.tm_out[k]; }
    __syncthreads();
    const float DCTnorm = 1.0f / (4 * TS * TS);
    const long long n = (long long)a.w * a.h;
#ifndef DETAIL_GATHER_LOOPS
    // Round 5: the nine candidate terms of a pixel as straight-line code.  Block row yc - 1 + s (s = 0, 1, 2; yc = y / 25) holds pixel row y at
    // its row i = y % 25 + 50 - 25 s: s = 1 and 2 always exist and contain it, s = 0 does when yc > 0 and y % 25 < 14; the same along x.  All
    // nine block values are loaded first -- a term that does not exist loads the (2, 2) term's address again, a line the lane has in flight
    // anyway -- and then summed in the reference's order (vblk, then hblk), a missing term by a select that leaves both sums as they are.
    // The nested loops this replaces skipped the missing terms with `continue`: each load sat in a divergent region of its own and the wave
    // waited for it before it issued the next (85 % of the kernel's wave cycles were such waits).
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(t / a.w), x = (int)(t - (long long)y * a.w);
        const int yc = y / OFF, xc = x / OFF, ry = y - yc * OFF, rx = x - xc * OFF;
        bool vs[3], hs[3];
        int iv[3], jh[3], vb[3], hb[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            vs[k] = k > 0 || (yc > 0 && ry < TS - 2 * OFF);
            hs[k] = k > 0 || (xc > 0 && rx < TS - 2 * OFF);
            iv[k] = vs[k] ? ry + 2 * OFF - OFF * k : ry;  vb[k] = vs[k] ? yc - 1 + k : yc + 1;
            jh[k] = hs[k] ? rx + 2 * OFF - OFF * k : rx;  hb[k] = hs[k] ? xc - 1 + k : xc + 1;
        }
        float blk[9], tmo[9], tmi[9];
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
            for (int t_ = 0; t_ < 3; ++t_) {
                const int m = iv[s_] * TS + jh[t_];
                blk[3 * s_ + t_] = a.blocks[((size_t)vb[s_] * a.numblox_W + hb[t_]) * TS * TS + m];
                tmo[3 * s_ + t_] = s_out[m];
                tmi[3 * s_ + t_] = s_in[m];
            }
        const float Lold = a.L[t];
        float Ldetail = 0.f, totwt = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
            for (int t_ = 0; t_ < 3; ++t_) {
                const bool v = vs[s_] && hs[t_];
                const float ld = Ldetail + tmo[3 * s_ + t_] * blk[3 * s_ + t_] * DCTnorm, tw = totwt + tmi[3 * s_ + t_] * tmo[3 * s_ + t_];
                Ldetail = v ? ld : Ldetail;
                totwt = v ? tw : totwt;
            }
        a.L[t] = Lold + Ldetail / totwt;
    }
#else
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(t / a.w), x = (int)(t - (long long)y * a.w);
        // blocks with top <= y < top + 64, top = (vblk - 1) * 25: at most three per axis
        const int vb0 = max(0, y / OFF + BLKRAD - 2), vb1 = min(a.numblox_H - 1, y / OFF + BLKRAD);
        const int hb0 = max(0, x / OFF + BLKRAD - 2), hb1 = min(a.numblox_W - 1, x / OFF + BLKRAD);
        float Ldetail = 0.f, totwt = 0.f;
        for (int vblk = vb0; vblk <= vb1; ++vblk) {
            const int i = y - (vblk - BLKRAD) * OFF;
            if (i < 0 || i >= TS) continue;
            for (int hblk = hb0; hblk <= hb1; ++hblk) {
                const int j = x - (hblk - BLKRAD) * OFF;
                if (j < 0 || j >= TS) continue;
                const float tmo = s_out[i * TS + j];
                Ldetail += tmo * a.blocks[((size_t)vblk * a.numblox_W + hblk) * TS * TS + i * TS + j] * DCTnorm;
                totwt += s_in[i * TS + j] * tmo;
            }
        }
        a.L[t] += Ldetail / totwt;
    }
#endif
}

hipError_t launch_detail_blocks(const DetailArgs &a, hipStream_t s)
{
    // blur_rad = max(1, int(3 / scale)), scale >= 1 (FTblockDN.cc:1499): 1, 2 or 3
    const dim3 grid(NXCD * ((a.numblox_W + NXCD - 1) / NXCD) * a.numblox_H);      // (an eighth of the block columns per XCD, padded)
    switch (a.blur_rad) {
    case 1: hipLaunchKernelGGL(detail_blocks_kernel<1>, grid, dim3(64), 0, s, a); break;
    case 2: hipLaunchKernelGGL(detail_blocks_kernel<2>, grid, dim3(64), 0, s, a); break;
    case 3: hipLaunchKernelGGL(detail_blocks_kernel<3>, grid, dim3(64), 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t launch_detail_gather(const DetailArgs &a, hipStream_t s)
{
    const long long n = (long long)a.w * a.h;
    const long long g = (n + 1023) / 1024;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long long cap = 2LL * cus;           // persistent: 32 KB of mask tables per 1024-thread workgroup, two per CU
    hipLaunchKernelGGL(detail_gather_kernel, dim3((int)(g < cap ? g : cap)), dim3(1024), 0, s, a);
    return hipGetLastError();
}

} // namespace artgpu
