// art_amd/csrc/pixelops.hip -- per-pixel stages around the demosaic, one lane per pixel,
// coalesced row accesses, HBM-bound (12 B in + 12 B out per pixel):
//   get_image_convert : RawImageSource::getImage (skip=1, tran=0; rawimagesource.cc:940-1025)
//                       fused with the matrix branch of colorSpaceConversion_ (L3184-3213;
//                       double accumulation, float store) -- the intermediate the reference
//                       stores between the two is the same float the fused kernel converts
//   convert_color_space: the matrix branch alone, in place
//   exposure           : ImProcFunctions::expcomp (ipexposure.cc:28-72), SSE lanes use
//                        _mm_max_ps(x,0), the W%4 tail uses std::max(x,0.f)
//   tone_std           : filmlike_clip (iptonecurve.cc:214-231 -> color.cc:6648-6690) followed by
//                        StandardToneCurve::Apply (curves.h:224-231,360-368; LUT.h:436-459)
#include <hip/hip_runtime.h>
#include "devmath.h"
#include "devsleef.h"
#include "kernels.h"

namespace artgpu {

namespace {
__device__ __forceinline__ float clipf(float a) { return std_max(0.f, std_min(a, 65535.f)); }

__device__ __forceinline__ void clip_rgb_tone(float &r, float &g, float &b, float L)
{
    const float r_ = r > L ? L : r;
    const float b_ = b > L ? L : b;
    const float g_ = b_ + ((r_ - b_) * (g - b) / (r - b));
    r = r_; g = g_; b = b_;
}
__device__ __forceinline__ void filmlike_clip_px(float &r, float &g, float &b, float L)
{
    if (r >= g) {
        if (g > b) clip_rgb_tone(r, g, b, L);
        else if (b > r) clip_rgb_tone(b, r, g, L);
        else if (b > g) clip_rgb_tone(r, b, g, L);
        else { r = r > L ? L : r; g = g > L ? L : g; b = g; }
    } else {
        if (r >= b) clip_rgb_tone(g, r, b, L);
        else if (b > g) clip_rgb_tone(b, g, r, L);
        else clip_rgb_tone(g, b, r, L);
    }
}
__device__ __forceinline__ float lutf(const float *__restrict__ data, int size, float index)
{
    const int maxs = size - 2, upper = size - 1;
    if (index < 0.f || !(index == index)) return data[0];
    if (index > (float)maxs) return data[upper];
    const int idx = (int)index;
    const float diff = index - (float)idx;
    const float p1 = data[idx];
    const float p2 = data[idx + 1] - p1;
    return p1 + p2 * diff;
}
} // namespace

__global__ void __launch_bounds__(256) get_image_convert_kernel(PixArgs a)
{
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        size_t si = (size_t)(a.sy1 + y) * a.src_stride + a.sx1 + x;
        float r, g, b;
        if (a.has_mul && a.skip > 1) {
            // skip x skip box sums, rows outer (rawimagesource.cc:944-957); the window is pulled inside at the right/bottom edge
            const int i = min(a.sy1 + a.skip * y, a.src_h - a.skip), jx = min(a.sx1 + a.skip * x, a.src_w - a.skip);
            r = 0.f; g = 0.f; b = 0.f;
            for (int m = 0; m < a.skip; ++m)
                for (int n = 0; n < a.skip; ++n) {
                    si = (size_t)(i + m) * a.src_stride + jx + n;
                    r += a.src[0][si]; g += a.src[1][si]; b += a.src[2][si];
                }
            r *= a.mul[0]; g *= a.mul[1]; b *= a.mul[2];
            if (a.do_clip) { r = clipf(r); g = clipf(g); b = clipf(b); }
        } else if (a.has_mul) {
            r = 0.f; g = 0.f; b = 0.f;
            r += a.src[0][si]; g += a.src[1][si]; b += a.src[2][si];
            r *= a.mul[0]; g *= a.mul[1]; b *= a.mul[2];
            if (a.do_clip) { r = clipf(r); g = clipf(g); b = clipf(b); }
        } else {
            r = a.src[0][si]; g = a.src[1][si]; b = a.src[2][si];
        }
        if (a.has_mat) {
            const double dr = r, dg = g, db = b;
            r = (float)(a.mat[0] * dr + a.mat[1] * dg + a.mat[2] * db);
            g = (float)(a.mat[3] * dr + a.mat[4] * dg + a.mat[5] * db);
            b = (float)(a.mat[6] * dr + a.mat[7] * dg + a.mat[8] * db);
        }
        const size_t di = (size_t)y * a.dst_stride + x;
        a.dst[0][di] = r; a.dst[1][di] = g; a.dst[2][di] = b;
    }
}

__global__ void __launch_bounds__(256) exposure_kernel(PixArgs a)
{
    const int wv = (a.w / 4) * 4;
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        const size_t di = (size_t)y * a.dst_stride + x;
        // the three loads first: the planes may alias as far as the compiler knows, so load - store - load - store ... was three
        // dependent memory round trips per pixel
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = a.dst[c][di];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float e = v[c] * a.exp_scale - a.black;
            float o = x < wv ? sse_max(e, 0.f) : std_max(e, 0.f);
            for (int k = 0; k < a.chain_n; ++k) {          // exposure steps that follow directly (ipdenoise.cc:1181-1184, then process STAGE_1)
                const float e2 = o * a.chain_scale[k] - a.chain_black[k];
                o = x < wv ? sse_max(e2, 0.f) : std_max(e2, 0.f);
            }
            a.dst[c][di] = o;
        }
    }
}

template <bool PC>
__global__ void __launch_bounds__(256) tone_std_kernel(PixArgs a)
{
    const float Lmax = 65535.f * a.whitept;
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        const size_t di = (size_t)y * a.dst_stride + x;
        float r = a.dst[0][di], g = a.dst[1][di], b = a.dst[2][di];
        if (a.do_clip) filmlike_clip_px(r, g, b, Lmax);
        if (a.lut) {
            r = (a.tail_kind && r > 65535.f) ? curve_tail<PC>(a.tail_kind, a.tail_y, a.tail_pc, r) : lutf(a.lut, 65536, std_max(r, 0.f));
            g = (a.tail_kind && g > 65535.f) ? curve_tail<PC>(a.tail_kind, a.tail_y, a.tail_pc, g) : lutf(a.lut, 65536, std_max(g, 0.f));
            b = (a.tail_kind && b > 65535.f) ? curve_tail<PC>(a.tail_kind, a.tail_y, a.tail_pc, b) : lutf(a.lut, 65536, std_max(b, 0.f));
        }
        a.dst[0][di] = r; a.dst[1][di] = g; a.dst[2][di] = b;
    }
}

// The same with the lower LUT_LDS_N entries of the curve resident in LDS (lutf_lookup_lds, devsleef.h; 159 KB of the CU's 160 KB): the 65536-entry LUT
// (256 KB) does not fit L1, so every lookup of the kernel above is an L2 line gather (6 per pixel, ~1.5 TB/s effective); linear
// scene data sit mostly in the lower part of the range, and those lookups become LDS reads.  Entries above come from L2 as before.
// One persistent workgroup of 1024 threads per CU, rows strided over the workgroups.
template <bool PC>
__global__ void __launch_bounds__(1024) tone_std_lds_kernel(PixArgs a)
{
    extern __shared__ float tone_lds[];
    lut_lds_fill(tone_lds, a.lut, 1024);
    const float Lmax = 65535.f * a.whitept;
    for (int yb = blockIdx.x; yb < a.h; yb += LDSK_ROWS * gridDim.x)
        for (int x0 = 0; x0 < a.w; x0 += LDSK_PX * 1024) {
            float r[LDSK_ROWS * LDSK_PX], g[LDSK_ROWS * LDSK_PX], b[LDSK_ROWS * LDSK_PX];
#pragma unroll
            for (int k = 0; k < LDSK_ROWS * LDSK_PX; ++k) {
                const int x = x0 + (k % LDSK_PX) * 1024 + (int)threadIdx.x, yr = yb + (k / LDSK_PX) * (int)gridDim.x, y = yr < a.h ? yr : a.h - 1;
                const size_t di = (size_t)y * a.dst_stride + (x < a.w ? x : a.w - 1);
                r[k] = a.dst[0][di]; g[k] = a.dst[1][di]; b[k] = a.dst[2][di];
            }
#pragma unroll
            for (int k = 0; k < LDSK_ROWS * LDSK_PX; ++k) {
                const int x = x0 + (k % LDSK_PX) * 1024 + (int)threadIdx.x, yr = yb + (k / LDSK_PX) * (int)gridDim.x, y = yr < a.h ? yr : a.h - 1;
                if (x >= a.w || yr >= a.h) continue;
                const size_t di = (size_t)y * a.dst_stride + x;
                float rr = r[k], gg = g[k], bb = b[k];
                if (a.do_clip) filmlike_clip_px(rr, gg, bb, Lmax);
                rr = (a.tail_kind && rr > 65535.f) ? curve_tail<PC>(a.tail_kind, a.tail_y, a.tail_pc, rr) : lutf_lookup_lds<true>(tone_lds, a.lut, 65536, std_max(rr, 0.f));
                gg = (a.tail_kind && gg > 65535.f) ? curve_tail<PC>(a.tail_kind, a.tail_y, a.tail_pc, gg) : lutf_lookup_lds<true>(tone_lds, a.lut, 65536, std_max(gg, 0.f));
                bb = (a.tail_kind && bb > 65535.f) ? curve_tail<PC>(a.tail_kind, a.tail_y, a.tail_pc, bb) : lutf_lookup_lds<true>(tone_lds, a.lut, 65536, std_max(bb, 0.f));
                a.dst[0][di] = rr; a.dst[1][di] = gg; a.dst[2][di] = bb;
            }
        }
}

// Imagefloat::setMode RGB -> YUV (imagefloat.cc:700-725: Y -> g plane, u = Y-b -> b plane, v = r-Y -> r plane) and
// YUV -> RGB (L779-804), float working-space row ws_[1][*]; in place.  a.do_clip: 0 = to YUV, 1 = to RGB.
__global__ void __launch_bounds__(256) yuv_mode_kernel(PixArgs a)
{
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        const size_t di = (size_t)y * a.dst_stride + x;
        if (!a.do_clip) {
            const float r = a.dst[0][di], g = a.dst[1][di], b = a.dst[2][di];
            const float Y = r * a.mul[0] + g * a.mul[1] + b * a.mul[2];
            a.dst[1][di] = Y; a.dst[2][di] = Y - b; a.dst[0][di] = r - Y;
        } else {
            const float Y = a.dst[1][di], u = a.dst[2][di], v = a.dst[0][di];
            const float b = Y - u, r = v + Y;
            const float g = (Y - r * a.mul[0] - b * a.mul[2]) / a.mul[1];
            float o[3] = {r, g, b};
            if (a.chain_n) {                                // the exposure steps that follow setMode(RGB) directly (exposure_kernel's forms)
                const bool vl = x < (a.w / 4) * 4;
                for (int k = 0; k < a.chain_n; ++k)
                    for (int c = 0; c < 3; ++c) {
                        const float e = o[c] * a.chain_scale[k] - a.chain_black[k];
                        o[c] = vl ? sse_max(e, 0.f) : std_max(e, 0.f);
                    }
            }
            a.dst[0][di] = o[0]; a.dst[1][di] = o[1]; a.dst[2][di] = o[2];
        }
    }
}

// RawImageSource::copyOriginalPixels without dark frame / flat field (rawData = (float)src->data) followed by scaleColors
// (rawimagesource.cc:2739-2760 Bayer, 2806-2813 X-Trans): val = max(0, raw - cblacksom[c4]) * scale_mul[c4]; chmax[c] = max.
__global__ void __launch_bounds__(256) scale_colors_kernel(ScaleArgs a)
{
    __shared__ int s_max[3];
    if (threadIdx.x < 3) s_max[threadIdx.x] = 0;
    __syncthreads();
    float m[3] = {0.f, 0.f, 0.f};
    FOR_IMAGE_XY(row, col, a.w, a.h) {
        const size_t si = (size_t)row * a.src_stride + col;
        float val = a.src_u16 ? (float)static_cast<const unsigned short *>(a.src)[si] : static_cast<const float *>(a.src)[si];
        const int c = a.cfa[(row % 6) * 6 + col % 6];
        const int c4 = (a.bayer && c == 1 && !(row & 1)) ? 3 : c;
        val = std_max(0.f, val - a.cblacksom[c4]) * a.scale_mul[c4];
        a.dst[(size_t)row * a.dst_stride + col] = val;
        m[c] = std_max(m[c], val);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) atomicMax(&s_max[c], __float_as_int(m[c]));     // non-negative floats order like their bit patterns
    __syncthreads();
    if (threadIdx.x < 3) atomicMax(&a.chmax_bits[threadIdx.x], s_max[threadIdx.x]);
}

// ImProcFunctions::channelMixer pixel loop (ipchmixer.cc:200-230): 4-lane groups clamp with vmaxf, the row tail with max()
__global__ void __launch_bounds__(256) channel_mixer_kernel(MixArgs a)
{
    const int wv = a.w - 3 > 0 ? ((a.w - 3 + 3) / 4) * 4 : 0;    // x < W-3 in steps of 4
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        const size_t di = (size_t)y * a.stride + x;
        const float r = a.dst[0][di], g = a.dst[1][di], b = a.dst[2][di];
        const float rmix = (r * a.m[0] + g * a.m[1] + b * a.m[2]);
        const float gmix = (r * a.m[3] + g * a.m[4] + b * a.m[5]);
        const float bmix = (r * a.m[6] + g * a.m[7] + b * a.m[8]);
        if (x < wv) { a.dst[0][di] = sse_max(rmix, 0.f); a.dst[1][di] = sse_max(gmix, 0.f); a.dst[2][di] = sse_max(bmix, 0.f); }
        else { a.dst[0][di] = std_max(rmix, 0.f); a.dst[1][di] = std_max(gmix, 0.f); a.dst[2][di] = std_max(bmix, 0.f); }
    }
}
// ImProcFunctions::rgbCurves pixel loop (iprgbcurves.cc:116-143): LUTf(65536, flags 0); 4-lane groups use the clamping vector
// lookup (LUT.h:349-377), the row tail the scalar one, which extrapolates (LUT.h:436-459)
__global__ void __launch_bounds__(256) rgb_curves_kernel(MixArgs a)
{
    const int wv = a.w - 3 > 0 ? ((a.w - 3 + 3) / 4) * 4 : 0;
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        const size_t di = (size_t)y * a.stride + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float *lut = a.lut[c];
            if (!lut) continue;
            const float v = a.dst[c][di];
            float r;
            if (x < wv) {
                const float clamped = sse_max(sse_min(65534.f, v), 0.f);
                const int idx = (int)clamped;
                const float diff = sse_max(sse_min(65535.f, v), 0.f) - (float)idx;
                r = intp(diff, lut[idx + 1], lut[idx]);
            } else {
                int idx = (int)v;
                if (v < 0.f || !(v == v)) idx = 0;
                else if (v > 65534.f) idx = 65534;
                const float diff = v - (float)idx, p1 = lut[idx], p2 = lut[idx + 1] - p1;
                r = p1 + p2 * diff;
            }
            a.dst[c][di] = r;
        }
    }
}

static int pix_grid(const PixArgs &a)
{
    const long long n = (long long)a.w * a.h;
    const long long g = (n + 255) / 256;
    return (int)(g < 16384 ? g : 16384);
}
hipError_t launch_get_image_convert(const PixArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(get_image_convert_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_exposure(const PixArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(exposure_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}
// ImProcFunctions::saturationVibrance (ipsaturation.cc:29-83): chroma = rgb - luminance (double working-space row), optional
// vibrance as a power law on |chroma| above 2^-16, then l + saturation * chroma floored at 2^-16
__device__ __forceinline__ float apply_vibrance_px(float x, float vib, float noise)
{
    const float ax = fabsf(x / 65535.f);
    if (ax > noise) {
        const float sgn = (float)((0.f < x) - (x < 0.f));
        return sgn * pow_F(ax, vib) * 65535.f;      // pow_F: sleef.h:1309-1313
    }
    return x;
}
__global__ void __launch_bounds__(256) saturation_vibrance_kernel(SatArgs a)
{
    const float noise = pow_F(2.f, -16.f);
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        const size_t di = (size_t)y * a.stride + x;
        const float r = a.dst[0][di], g = a.dst[1][di], b = a.dst[2][di];
        const float l = (float)(r * a.ws1[0] + g * a.ws1[1] + b * a.ws1[2]);
        float rl = r - l, gl = g - l, bl = b - l;
        if (a.vib) {
            rl = apply_vibrance_px(rl, a.vibrance, noise);
            gl = apply_vibrance_px(gl, a.vibrance, noise);
            bl = apply_vibrance_px(bl, a.vibrance, noise);
        }
        a.dst[0][di] = std_max(l + a.saturation * rl, noise);
        a.dst[1][di] = std_max(l + a.saturation * gl, noise);
        a.dst[2][di] = std_max(l + a.saturation * bl, noise);
    }
}
hipError_t launch_saturation_vibrance(const SatArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(saturation_vibrance_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ARTOutputProfile::operator()(const Imagefloat*, Imagefloat*) (iprgb2out.cc:152-172): /65535, 3x3 float matrix (accumulated from 0
// in column order like linalgebra.h:227-239), per channel the TRC from a LUT for values <= 1 (LUTf, clipped both ways) or, in
// linear mode, the value itself; x65535.  A value that would need ARTOutputProfile::eval's lcms2 / libm curve is counted.
__global__ void __launch_bounds__(256) rgb2out_matrix_kernel(OutArgs a)
{
    const float factor = (float)(a.lutsz - 1);
    int bad = 0;
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        const size_t si = (size_t)y * a.src_stride + x, di = (size_t)y * a.dst_stride + x;
        const float r = a.src[0][si] / 65535.f, g = a.src[1][si] / 65535.f, b = a.src[2][si] / 65535.f;
        float v[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float acc = 0.f;
            acc += a.m[3 * i] * r; acc += a.m[3 * i + 1] * g; acc += a.m[3 * i + 2] * b;
            if (a.lutsz > 0 && acc <= 1.f) acc = lutf(a.lut, a.lutsz, acc * factor);
            else if (!a.linear) ++bad;
            v[i] = acc;
        }
        a.dst[0][di] = v[0] * 65535.f; a.dst[1][di] = v[1] * 65535.f; a.dst[2][di] = v[2] * 65535.f;
    }
    if (bad) atomicAdd(a.unsupported, bad);
}
// Imagefloat::getScanline for every row (imagefloat.cc:125-170): interleaved RGB, 8/16-bit integer, half or float
__device__ __forceinline__ unsigned scan_elem(float v, int bps, int is_float)
{
    if (is_float) return bps == 32 ? __float_as_uint(v / 65535.f) : (unsigned)float_to_half_dng(v / 65535.f);
    const unsigned short q = (unsigned short)clipf(v);                       // CLIP, then the implicit float -> uint16 conversion
    return bps == 16 ? (unsigned)q : (unsigned)(unsigned char)((((int)q + 128) - (((int)q + 128) >> 8)) >> 8);     // uint16ToUint8Rounded
}
__global__ void __launch_bounds__(256) scanlines_kernel(OutArgs a)
{
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        const size_t si = (size_t)y * a.src_stride + x;
        unsigned char *row = a.out + (size_t)y * a.out_stride_bytes;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned e = scan_elem(a.src[c][si], a.bps, a.is_float);
            if (a.bps == 32) reinterpret_cast<unsigned *>(row)[3 * x + c] = e;
            else if (a.bps == 16) reinterpret_cast<unsigned short *>(row)[3 * x + c] = (unsigned short)e;
            else row[3 * x + c] = (unsigned char)e;
        }
    }
}
// The same scanlines written straight into the caller's pinned host buffer (artgpu_batch_run_io): a FEW persistent workgroups, each turning
// 2048 pixels of a row into the writers' bytes in LDS and sending them off as 16-byte pieces, 1 KB contiguous per wave and store -- the
// stores are posted, so a workgroup goes on with its next chunk while they cross PCIe.  With a grid of a few workgroups the download
// takes the time PCIe takes (268 MB of 16-bit scanlines in ~5 ms) on a few CUs, beside the next frame's kernels; the runtime's own
// device -> host copy is a kernel too, but one that fills every CU's wave slots with waves waiting on PCIe.
// Rows and the buffer are 16-byte aligned (the launcher checks); the last piece of a row is written byte by byte.
constexpr int SCAN_T = 1024, SCAN_CHUNK = 2 * SCAN_T;
typedef unsigned scan_u4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(SCAN_T) scanlines_host_kernel(OutArgs a)
{
    // two chunk buffers: the barrier between converting a chunk and sending it off orders LDS traffic only (a __syncthreads() would also wait
    // for the previous chunk's stores to come back across PCIe), and the planes' values of the NEXT chunk are fetched before this one is
    // sent off, so that a workgroup's time per chunk is not a memory round trip plus a PCIe round trip
    __shared__ __attribute__((aligned(16))) unsigned char bufs[2][SCAN_CHUNK * 3 * 4];
    const int esz = a.bps / 8, cpr = (a.w + SCAN_CHUNK - 1) / SCAN_CHUNK;
    const long long nchunks = (long long)cpr * a.h;
    const int tid = threadIdx.x;
    float v[2][3];
    auto fetch = [&](long long ch) {
        const int y = (int)(ch / cpr), x0 = (int)(ch - (long long)y * cpr) * SCAN_CHUNK;
        const int npx = min(SCAN_CHUNK, a.w - x0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int xl = min(tid + SCAN_T * q, npx - 1);
            const size_t si = (size_t)y * a.src_stride + x0 + xl;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[q][c] = a.src[c][si];
        }
    };
    if ((long long)blockIdx.x < nchunks) fetch(blockIdx.x);
    int par = 0;
    for (long long ch = blockIdx.x; ch < nchunks; ch += gridDim.x, par ^= 1) {
        unsigned char *buf = bufs[par];
        const int y = (int)(ch / cpr), x0 = (int)(ch - (long long)y * cpr) * SCAN_CHUNK;
        const int npx = min(SCAN_CHUNK, a.w - x0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int xl = tid + SCAN_T * q;
            if (xl >= npx) break;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const unsigned e = scan_elem(v[q][c], a.bps, a.is_float);
                if (a.bps == 32) reinterpret_cast<unsigned *>(buf)[3 * xl + c] = e;
                else if (a.bps == 16) reinterpret_cast<unsigned short *>(buf)[3 * xl + c] = (unsigned short)e;
                else buf[3 * xl + c] = (unsigned char)e;
            }
        }
        if (ch + gridDim.x < nchunks) fetch(ch + gridDim.x);
        // (one barrier per chunk is enough with two buffers: a thread that is a chunk ahead writes the OTHER buffer, and it cannot be two
        // chunks ahead without having passed the barrier of the chunk in between, which every reader of this buffer's previous contents
        // reaches only after its reads)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        unsigned char *dst = a.out + (size_t)y * a.out_stride_bytes + (size_t)x0 * 3 * esz;
        const int nbytes = npx * 3 * esz, whole = nbytes & ~15;
        for (int o = tid * 16; o < whole; o += SCAN_T * 16)
            *reinterpret_cast<scan_u4 *>(dst + o) = *reinterpret_cast<const scan_u4 *>(buf + o);
        if (tid < nbytes - whole) dst[whole + tid] = buf[whole + tid];
    }
}
hipError_t launch_rgb2out_matrix(const OutArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(rgb2out_matrix_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_scanlines(const OutArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(scanlines_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}
bool scanlines_host_ok(const OutArgs &a) { return (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 && (a.out_stride_bytes & 15) == 0; }
hipError_t launch_scanlines_host(const OutArgs &a, int workgroups, hipStream_t s)
{
    const long long nchunks = (long long)((a.w + SCAN_CHUNK - 1) / SCAN_CHUNK) * a.h;
    const int g = (int)(nchunks < workgroups ? (nchunks ? nchunks : 1) : (workgroups < 1 ? 1 : workgroups));
    hipLaunchKernelGGL(scanlines_host_kernel, dim3(g), dim3(SCAN_T), 0, s, a);
    return hipGetLastError();
}

static unsigned mix_grid(const MixArgs &a) { long long g = ((long long)a.w * a.h + 255) / 256; return (unsigned)(g < 16384 ? (g ? g : 1) : 16384); }
hipError_t launch_channel_mixer(const MixArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(channel_mixer_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_rgb_curves(const MixArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(rgb_curves_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_scale_colors(const ScaleArgs &a, hipStream_t s)
{
    // a bounded grid: every workgroup ends with three atomic maxima on the same three words, and one workgroup per row segment of a 45 MP frame
    // (350 000 of them) spent 2.2 ms queueing up for those words; 256 rows of workgroups walk the frame instead (~0.1 ms)
    dim3 g = image_grid(a.w, a.h);
    if (g.y > 256) g.y = 256;
    hipLaunchKernelGGL(scale_colors_kernel, g, dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_yuv_mode(const PixArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(yuv_mode_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_tone_std(const PixArgs &a, hipStream_t s)
{
    // large frames with a curve: the LDS-resident variant (the curve pointer is 16-byte aligned: pool or hipMalloc memory)
    if (a.lut && (long long)a.w * a.h >= (1 << 22) && (reinterpret_cast<uintptr_t>(a.lut) & 15) == 0 && !a.no_lds_lut && device_block_fits(LUT_LDS_N * (int)sizeof(float), 1024)) {
        const size_t lds = (size_t)LUT_LDS_N * sizeof(float);
        const bool pc = a.tail_kind == 4;
        hipError_t e = hipFuncSetAttribute(pc ? reinterpret_cast<const void *>(tone_std_lds_kernel<true>) : reinterpret_cast<const void *>(tone_std_lds_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (a.cu_reserve > 0) cus = cus - a.cu_reserve > 1 ? cus - a.cu_reserve : 1;
        if (pc) hipLaunchKernelGGL(tone_std_lds_kernel<true>, dim3(cus < a.h ? cus : a.h), dim3(1024), lds, s, a);
        else hipLaunchKernelGGL(tone_std_lds_kernel<false>, dim3(cus < a.h ? cus : a.h), dim3(1024), lds, s, a);
        return hipGetLastError();
    }
    if (a.tail_kind == 4) hipLaunchKernelGGL(tone_std_kernel<true>, image_grid(a.w, a.h), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(tone_std_kernel<false>, image_grid(a.w, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}

} // namespace artgpu
