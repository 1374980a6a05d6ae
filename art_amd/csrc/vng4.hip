// art_amd/csrc/vng4.hip -- RawImageSource::vng4_demosaic on gfx950 (reference: rtengine/vng4_demosaic_RT.cc:32-397; dcraw's VNG with
// the four-colour CFA description).
//
// The reference walks a table ("code") per (row & 7, col & 1) cell: pairs of same-colour neighbours whose absolute difference, times a
// power-of-two weight, is added to one or two of eight directional gradients -- in table order, which is the fp32 summation order --
// then averages the neighbours whose gradient is below min + max / 2.  The tables are built on the host exactly like the reference
// builds them (same terms, same skips) and interpreted per pixel here; every pixel is independent, so the three passes are plain grids:
//   vng4_linear   the four-colour image (16 B/px): native sample + the 3x3 weighted means of the other three colours
//   vng4_green    gradients + thresholded average -> green
//   vng4_redblue  linear colour-difference interpolation of red / blue against that green (rows / columns 3 .. n-4)
// and border_interpolate2(3) (border.hip).  HBM: 4 + 16 B/px, 16 x ~25 taps from L2 + 4, 12 + 8 B/px.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <string.h>
#include "devmath.h"
#include "kernels.h"

namespace artgpu {
namespace {

__global__ void __launch_bounds__(256) vng4_linear_kernel(Vng4Args a)
{
    const int W = a.w, H = a.h;
    FOR_IMAGE_XY(row, col, W, H) {
        const float *r = a.raw + (size_t)row * a.raw_stride + col;
        float pix[4] = {0.f, 0.f, 0.f, 0.f};
        const unsigned own = fc(a.prefilters, row, col);
        pix[own] = r[0];
        if (row >= 1 && row < H - 1 && col >= 1 && col < W - 1) {
            float sum[4] = {0.f, 0.f, 0.f, 0.f}, wsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int y = -1; y <= 1; y++)
#pragma unroll
                for (int x = -1; x <= 1; x++) {
                    const int shift = (y == 0) + (x == 0);
                    if (shift == 2) continue;
                    const unsigned color = fc(a.prefilters, row + y, col + x);
                    const float v = r[(long long)y * (long long)a.raw_stride + x] * (float)(1 << shift);
#pragma unroll
                    for (unsigned c = 0; c < 4; ++c)
                        if (c == color) { sum[c] += v; wsum[c] += (float)(1 << shift); }
                }
#pragma unroll
            for (unsigned c = 0; c < 4; c++)
                if (c != own) pix[c] = sum[c] * (1.f / wsum[c]);
        }
        reinterpret_cast<float4 *>(a.image)[(size_t)row * W + col] = make_float4(pix[0], pix[1], pix[2], pix[3]);
    }
}

__device__ __forceinline__ float min8(const float *g) { float m = g[0]; for (int k = 1; k < 8; ++k) m = std_min(m, g[k]); return m; }
__device__ __forceinline__ float max8(const float *g) { float m = g[0]; for (int k = 1; k < 8; ++k) m = std_max(m, g[k]); return m; }

__global__ void __launch_bounds__(256) vng4_green_kernel(Vng4Args a)
{
    const int W = a.w, H = a.h;
    FOR_IMAGE_XY(row, col, W, H) {
        if (row < 2 || row >= H - 2 || col < 2 || col >= W - 2) continue;
        const float *pix = a.image + ((size_t)row * W + col) * 4;
        int color = (int)fc(a.prefilters, row, col);
        const int *ip = a.code + ((row & 7) * 2 + (col & 1)) * VNG4_CODE_INTS;
        float gval[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        while (ip[0] != INT_MAX) {
            const float diff = fabsf(pix[ip[0]] - pix[ip[1]]) * __int_as_float(ip[2]);
            const int g0 = ip[3];
            ip += 5;
            const int g1 = ip[-1];
#pragma unroll
            for (int g = 0; g < 8; ++g) if (g == g0 || g == g1) gval[g] += diff;      // g1 == -1: no second gradient
            if (g1 != -1) ip++;
        }
        ip++;
        const float thold = min8(gval) + max8(gval) * 0.5f;
        float sum0 = 0.f, sum1 = 0.f;
        const float greenval = pix[color];
        int num = 0;
        if (color & 1) {
            color ^= 2;
#pragma unroll
            for (int g = 0; g < 8; g++, ip += 2)
                if (gval[g] <= thold) {
                    if (ip[1]) sum0 += greenval + pix[ip[1]];
                    sum1 += pix[ip[0] + color];
                    num++;
                }
            sum0 *= 0.5f;
        } else {
#pragma unroll
            for (int g = 0; g < 8; g++, ip += 2)
                if (gval[g] <= thold) {
                    if (ip[1]) sum0 += greenval + pix[ip[1]];
                    sum1 += pix[ip[0] + 1] + pix[ip[0] + 3];
                    num++;
                }
        }
        a.green[(size_t)row * a.out_stride + col] = std_max(0.f, greenval + (sum1 - sum0) / (float)(2 * num));
    }
}

// vng4interpolate_row_redblue (L32-57)
__global__ void __launch_bounds__(256) vng4_redblue_kernel(Vng4Args a)
{
    const int W = a.w, H = a.h;
    FOR_IMAGE_XY(i, j, W, H) {
        if (i < 3 || i >= H - 3 || j < 3 || j >= W - 3) continue;
        const bool swap = fc(a.filters, i, 0) == 2 || fc(a.filters, i, 1) == 2;
        float *ar = (swap ? a.blue : a.red) + (size_t)i * a.out_stride, *ab = (swap ? a.red : a.blue) + (size_t)i * a.out_stride;
        const float *cg = a.green + (size_t)i * a.out_stride, *pg = cg - a.out_stride, *ng = cg + a.out_stride;
        const float *r = a.raw + (size_t)i * a.raw_stride;
        const long long rs = (long long)a.raw_stride;
        if (fc(a.filters, i, j) != 1) {
            ar[j] = r[j];
            float rb = (r[-rs + j - 1] - pg[j - 1] + r[rs + j - 1] - ng[j - 1]);
            rb += (r[-rs + j + 1] - pg[j + 1] + r[rs + j + 1] - ng[j + 1]);
            ab[j] = std_max(0.f, cg[j] + rb * 0.25f);
        } else {
            ar[j] = std_max(0.f, cg[j] + (r[j - 1] - cg[j - 1] + r[j + 1] - cg[j + 1]) / 2);
            ab[j] = std_max(0.f, cg[j] + (r[-rs + j] - pg[j] + r[rs + j] - ng[j]) / 2);
        }
    }
}

// dual_demosaic_RT.cc:134-147: all three channels, every pixel
__global__ void __launch_bounds__(256) dual_blend_planes_kernel(DualArgs a, const float *r2, const float *g2, const float *b2, size_t s2)
{
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        const size_t o = (size_t)y * a.stride + x, q = (size_t)y * s2 + x;
        const float bl = a.blend[(size_t)y * a.w + x];
        a.rgb[0][o] = intp(bl, a.rgb[0][o], r2[q]);
        a.rgb[1][o] = intp(bl, a.rgb[1][o], g2[q]);
        a.rgb[2][o] = intp(bl, a.rgb[2][o], b2[q]);
    }
}

const signed short vng_terms[] = {
    -2, -2, +0, -1, 0, 0x01, -2, -2, +0, +0, 1, 0x01, -2, -1, -1, +0, 0, 0x01,
    -2, -1, +0, -1, 0, 0x02, -2, -1, +0, +0, 0, 0x03, -2, -1, +0, +1, 1, 0x01,
    -2, +0, +0, -1, 0, 0x06, -2, +0, +0, +0, 1, 0x02, -2, +0, +0, +1, 0, 0x03,
    -2, +1, -1, +0, 0, 0x04, -2, +1, +0, -1, 1, 0x04, -2, +1, +0, +0, 0, 0x06,
    -2, +1, +0, +1, 0, 0x02, -2, +2, +0, +0, 1, 0x04, -2, +2, +0, +1, 0, 0x04,
    -1, -2, -1, +0, 0, 0x80, -1, -2, +0, -1, 0, 0x01, -1, -2, +1, -1, 0, 0x01,
    -1, -2, +1, +0, 1, 0x01, -1, -1, -1, +1, 0, 0x88, -1, -1, +1, -2, 0, 0x40,
    -1, -1, +1, -1, 0, 0x22, -1, -1, +1, +0, 0, 0x33, -1, -1, +1, +1, 1, 0x11,
    -1, +0, -1, +2, 0, 0x08, -1, +0, +0, -1, 0, 0x44, -1, +0, +0, +1, 0, 0x11,
    -1, +0, +1, -2, 1, 0x40, -1, +0, +1, -1, 0, 0x66, -1, +0, +1, +0, 1, 0x22,
    -1, +0, +1, +1, 0, 0x33, -1, +0, +1, +2, 1, 0x10, -1, +1, +1, -1, 1, 0x44,
    -1, +1, +1, +0, 0, 0x66, -1, +1, +1, +1, 0, 0x22, -1, +1, +1, +2, 0, 0x10,
    -1, +2, +0, +1, 0, 0x04, -1, +2, +1, +0, 1, 0x04, -1, +2, +1, +1, 0, 0x04,
    +0, -2, +0, +0, 1, 0x80, +0, -1, +0, +1, 1, 0x88, +0, -1, +1, -2, 0, 0x40,
    +0, -1, +1, +0, 0, 0x11, +0, -1, +2, -2, 0, 0x40, +0, -1, +2, -1, 0, 0x20,
    +0, -1, +2, +0, 0, 0x30, +0, -1, +2, +1, 1, 0x10, +0, +0, +0, +2, 1, 0x08,
    +0, +0, +2, -2, 1, 0x40, +0, +0, +2, -1, 0, 0x60, +0, +0, +2, +0, 1, 0x20,
    +0, +0, +2, +1, 0, 0x30, +0, +0, +2, +2, 1, 0x10, +0, +1, +1, +0, 0, 0x44,
    +0, +1, +1, +2, 0, 0x10, +0, +1, +2, -1, 1, 0x40, +0, +1, +2, +0, 0, 0x60,
    +0, +1, +2, +1, 0, 0x20, +0, +1, +2, +2, 0, 0x10, +1, -2, +1, +0, 0, 0x80,
    +1, -1, +1, +1, 0, 0x88, +1, +0, +1, +2, 0, 0x08, +1, +0, +2, -1, 0, 0x40,
    +1, +0, +2, +1, 0, 0x10
};
const signed short vng_chood[] = {-1, -1, -1, 0, -1, +1, 0, +1, +1, +1, +1, 0, +1, -1, 0, -1};
inline unsigned hfc(unsigned f, int row, int col) { return (f >> ((((row << 1) & 14) + (col & 1)) << 1)) & 3u; }

} // namespace

// dcraw identify(): the second green of a three-colour pattern becomes colour 3 (what RawImage::set_prefilters keeps in `prefilters`)
unsigned vng4_prefilters(unsigned filters)
{
    return filters | ((((filters >> 2) & 0x22222222u) | ((filters << 2) & 0x88888888u)) & (filters << 1));
}
// the 8 x 2 code tables (vng4_demosaic_RT.cc:224-282), VNG4_CODE_INTS ints each; offsets in floats of the 4-float-per-pixel image
void vng4_build_code(unsigned pf, int width, int *codes)
{
    for (int row = 0; row < 8; row++)
        for (int col = 0; col < 2; col++) {
            int *ip = codes + (row * 2 + col) * VNG4_CODE_INTS;
            const signed short *cp = vng_terms;
            for (int t = 0; t < 64; t++) {
                const int y1 = *cp++, x1 = *cp++, y2 = *cp++, x2 = *cp++, weight = *cp++, grads = *cp++;
                const unsigned color = hfc(pf, row + y1, col + x1);
                if (hfc(pf, row + y2, col + x2) != color) continue;
                const int diag = (hfc(pf, row, col + 1) == color && hfc(pf, row + 1, col) == color) ? 2 : 1;
                if (abs(y1 - y2) == diag && abs(x1 - x2) == diag) continue;
                *ip++ = (y1 * width + x1) * 4 + (int)color;
                *ip++ = (y2 * width + x2) * 4 + (int)color;
                { const float w = (float)(1 << weight); int b; memcpy(&b, &w, 4); *ip++ = b; }
                for (int g = 0; g < 8; g++) if (grads & (1 << g)) *ip++ = g;
                *ip++ = -1;
            }
            *ip++ = INT_MAX;
            cp = vng_chood;
            for (int g = 0; g < 8; g++) {
                const int y = *cp++, x = *cp++;
                *ip++ = (y * width + x) * 4;
                const unsigned color = hfc(pf, row, col);
                *ip++ = (hfc(pf, row + y, col + x) != color && hfc(pf, row + y * 2, col + x * 2) == color) ? (y * width + x) * 8 + (int)color : 0;
            }
        }
}

hipError_t launch_vng4(const Vng4Args &a, hipStream_t s)
{
    hipLaunchKernelGGL(vng4_linear_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    hipLaunchKernelGGL(vng4_green_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    hipLaunchKernelGGL(vng4_redblue_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_dual_blend_planes(const DualArgs &a, const float *r2, const float *g2, const float *b2, size_t s2, hipStream_t s)
{
    hipLaunchKernelGGL(dual_blend_planes_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a, r2, g2, b2, s2);
    return hipGetLastError();
}

} // namespace artgpu
