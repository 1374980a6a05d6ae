// art_amd/csrc/border.hip -- border_interpolate2 (reference: rtengine/demosaic_algos.cc:200-353).
//
// One lane per frame pixel.  3x3 neighbourhood clipped to the image, per-colour fp32 sums
// accumulated in the reference's raster order (i1 outer, j1 inner).  The four strips the
// reference visits (left, right: all rows; top, bottom: columns [bord, W-bord)) are disjoint
// for W > 2*bord, which artgpu_border_interpolate2 requires.
#include <hip/hip_runtime.h>
#include "devmath.h"
#include "kernels.h"

namespace artgpu {

__global__ void __launch_bounds__(256)
border_interpolate2_kernel(BorderArgs a)
{
    const int W = a.W, H = a.H, bord = a.bord;
    const long long nside = 2LL * bord * H;               // left + right strips
    const long long ninner = (long long)(W - 2 * bord);   // columns of the top/bottom strips
    const long long total = nside + 2LL * bord * ninner;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        int i, j;
        if (t < nside) {
            i = (int)(t / (2 * bord));
            const int k = (int)(t - (long long)i * 2 * bord);
            j = k < bord ? k : W - 2 * bord + k;
        } else {
            const long long u = t - nside;
            const int k = (int)(u / ninner);
            j = bord + (int)(u - k * ninner);
            i = k < bord ? k : H - 2 * bord + k;
        }
        float sum[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i1 = i - 1; i1 < i + 2; i1++)
            for (int j1 = j - 1; j1 < j + 2; j1++)
                if (i1 > -1 && i1 < H && j1 > -1 && j1 < W) {
                    const float v = a.raw[(size_t)i1 * a.raw_stride + j1];
                    const unsigned c = fc(a.filters, i1, j1);
                    // select-based accumulation keeps sum[] in registers
                    sum[0] = c == 0 ? sum[0] + v : sum[0];
                    sum[1] = c == 1 ? sum[1] + v : sum[1];
                    sum[2] = c == 2 ? sum[2] + v : sum[2];
                    sum[3] = c == 0 ? sum[3] + 1.f : sum[3];
                    sum[4] = c == 1 ? sum[4] + 1.f : sum[4];
                    sum[5] = c == 2 ? sum[5] + 1.f : sum[5];
                }
        const unsigned c = fc(a.filters, i, j);
        const float v = a.raw[(size_t)i * a.raw_stride + j];
        const size_t o = (size_t)i * a.out_stride + j;
        a.red[o] = c == 0 ? v : sum[0] / sum[3];
        a.green[o] = c == 1 ? v : sum[1] / sum[4];
        a.blue[o] = c == 2 ? v : sum[2] / sum[5];
    }
}

hipError_t launch_border_interpolate2(const BorderArgs &a, int grid, hipStream_t stream)
{
    hipLaunchKernelGGL(border_interpolate2_kernel, dim3(grid), dim3(256), 0, stream, a);
    return hipGetLastError();
}

} // namespace artgpu
