// tonecurve.hip -- NEUTRAL tone-curve mode (ART's default): NeutralToneCurve::BatchApply with basecurve == nullptr
// (rtengine/curves.cc:893-1038), one thread per pixel.  Jzazbz via the host-built PQ LUTs (color.cc:6706-6742);
// super-white LMS (> 1) falls back to powf per pixel like the reference (device powf there: tolerance).
#include "kernels.h"
#include "devmath.h"
#include "devsleef.h"

namespace artgpu {

namespace {

__device__ __forceinline__ float dev_PQ(float X)
{
    X = std_max(X, 1e-10f);
    const float XX = powf(X * 1e-4f, 0.1593017578125f);
    return powf((0.8359375f + 18.8515625f * XX) / (1 + 18.6875f * XX), 134.034375f);
}
__device__ __forceinline__ float dev_PQ_inv(float X)
{
    X = std_max(X, 1e-10f);
    const float XX = powf(X, 7.460772656268214e-03f);
    return 1e4f * powf((0.8359375f - XX) / (18.6875f * XX - 18.8515625f), 6.277394636015326f);
}
// LUTf::operator[](float), flags 0, index >= 0 here
__device__ __forceinline__ float lut_noclip(const float *__restrict__ data, float index)
{
    int idx = (int)index;
    if (index < 0.f || !(index == index)) idx = 0;
    else if (index > 65534.f) idx = 65534;
    const float diff = index - (float)idx;
    const float p1 = data[idx], p2 = data[idx + 1] - p1;
    return p1 + p2 * diff;
}
// the forward PQ table: six lookups per pixel, the hottest of the kernel's three 256 KB tables.  `lds` != nullptr: entries
// [0, LUT_LDS_N) are resident in LDS (the persistent launch shape below), the rest and the other two tables come from L2.
struct PqTab { const float *g; const float *lds; };
__device__ __forceinline__ float get_pq(const PqTab pq, float x)
{
    if (!(x >= 0.f && x <= 1.f)) return dev_PQ(x);
    if (!pq.lds) return lut_noclip(pq.g, x * 65535.f);
    const float index = x * 65535.f;
    int idx = (int)index;
    if (index > 65534.f) idx = 65534;
    const float diff = index - (float)idx;
    float p1, q;
    if (idx + 1 < LUT_LDS_N) { lds_cfloat *l = (lds_cfloat *)pq.lds; p1 = l[idx]; q = l[idx + 1]; }     // (LDS-qualified: ds_read, not flat_load)
    else { p1 = pq.g[idx]; q = pq.g[idx + 1]; }
    const float p2 = q - p1;
    return p1 + p2 * diff;
}
__device__ __forceinline__ float get_pq_inv(const float *__restrict__ pqi, float x) { return (x >= 0.f && x <= 1.f) ? lut_noclip(pqi, x * 65535.f) : dev_PQ_inv(x); }

// dot_product(Mat33, Vec3) (linalgebra.h:226-239): accumulates from 0
__device__ __forceinline__ void mat_vec(const float *m, const float v[3], float r[3])
{
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float acc = 0;
        acc += m[3 * i + 0] * v[0];
        acc += m[3 * i + 1] * v[1];
        acc += m[3 * i + 2] * v[2];
        r[i] = acc;
    }
}
__device__ __forceinline__ void rgb2jzczhz(const PqTab pq, float R, float G, float B, float &Jz, float &cz, float &hz, const float *ws)
{
    const float D[9] = {0.9555766f, -0.0230393f, 0.0631636f, -0.0282895f, 1.0099416f, 0.0210077f, 0.0122982f, -0.0204830f, 1.3299098f};
    float v[3] = {ws[0] * R + ws[1] * G + ws[2] * B, ws[3] * R + ws[4] * G + ws[5] * B, ws[6] * R + ws[7] * G + ws[8] * B}, d[3];
    mat_vec(D, v, d);
    const float X = d[0], Y = d[1], Z = d[2];
    const float Lp = get_pq(pq, 0.674207838f * X + 0.382799340f * Y - 0.047570458f * Z);
    const float Mp = get_pq(pq, 0.149284160f * X + 0.739628340f * Y + 0.083327300f * Z);
    const float Sp = get_pq(pq, 0.070941080f * X + 0.174768000f * Y + 0.670970020f * Z);
    const float Iz = 0.5f * (Lp + Mp);
    const float az = 3.524000f * Lp - 4.066708f * Mp + 0.542708f * Sp;
    const float bz = 0.199076f * Lp + 1.096799f * Mp - 1.295875f * Sp;
    Jz = (0.44f * Iz) / (1.f - 0.56f * Iz) - 1.6295499532821566e-11f;
    cz = sqrtf(bz * bz + az * az);
    hz = xatan2f_s(bz, az);
}
__device__ __forceinline__ void jzczhz2rgb(const float *__restrict__ pqi, float Jz, float cz, float hz, float &R, float &G, float &B, const float *iws)
{
    const float D[9] = {1.0478112f, 0.0228866f, -0.0501270f, 0.0295424f, 0.9904844f, -0.0170491f, -0.0092345f, 0.0150436f, 0.7521316f};
    float sn, cs;
    xsincosf_v(hz, sn, cs);
    const float bz = cz * sn, az = cz * cs;
    Jz = Jz + 1.6295499532821566e-11f;
    const float Iz = Jz / (0.44f + 0.56f * Jz);
    const float L = get_pq_inv(pqi, Iz + 1.386050432715393e-1f * az + 5.804731615611869e-2f * bz);
    const float M = get_pq_inv(pqi, Iz - 1.386050432715393e-1f * az - 5.804731615611891e-2f * bz);
    const float S = get_pq_inv(pqi, Iz - 9.601924202631895e-2f * az - 8.118918960560390e-1f * bz);
    float v[3], d[3];
    v[0] = +1.661373055774069e+00f * L - 9.145230923250668e-01f * M + 2.313620767186147e-01f * S;
    v[1] = -3.250758740427037e-01f * L + 1.571847038366936e+00f * M - 2.182538318672940e-01f * S;
    v[2] = -9.098281098284756e-02f * L - 3.127282905230740e-01f * M + 1.522766561305260e+00f * S;
    mat_vec(D, v, d);
    R = iws[0] * d[0] + iws[1] * d[1] + iws[2] * d[2];
    G = iws[3] * d[0] + iws[4] * d[1] + iws[5] * d[2];
    B = iws[6] * d[0] + iws[7] * d[1] + iws[8] * d[2];
}
// Color::filmlike_clip (color.cc:6650-6688)
__device__ __forceinline__ void clip_tone(float &r, float &g, float &b, const float L)
{
    const float r_ = r > L ? L : r;
    const float b_ = b > L ? L : b;
    const float g_ = b_ + ((r_ - b_) * (g - b) / (r - b));
    r = r_; g = g_; b = b_;
}
__device__ __forceinline__ void filmlike_clip_dev(float &r, float &g, float &b, float L)
{
    if (r >= g) {
        if (g > b) clip_tone(r, g, b, L);
        else if (b > r) clip_tone(b, r, g, L);
        else if (b > g) clip_tone(r, b, g, L);
        else { r = r > L ? L : r; g = g > L ? L : g; b = g; }
    } else {
        if (r >= b) clip_tone(g, r, b, L);
        else if (b > g) clip_tone(b, g, r, L);
        else clip_tone(g, b, r, L);
    }
}
__device__ __forceinline__ float gauss(float x, float b, float c) { return xexpf_s(-sqr(x - b) / (2 * sqr(c))); }

} // namespace

// NeutralToneCurve::ApplyState ctor, hue anchors (curves.cc:880-886) with hws = xyz_rec2020
__global__ void neutral_hues_kernel(NeutralArgs a)
{
    if (threadIdx.x || blockIdx.x) return;
    const float hws[9] = {0.6734241f, 0.1656411f, 0.1251286f, 0.2790177f, 0.6753402f, 0.0456377f, -0.0019300f, 0.0299784f, 0.7973330f};
    const float c[4][3] = {{1, 0, 0}, {0, 0, 1}, {1, 1, 0}, {1, 0.5f, 0}};
    for (int k = 0; k < 4; ++k) {
        float j, ch, hz;
        rgb2jzczhz(PqTab{a.pq, nullptr}, c[k][0], c[k][1], c[k][2], j, ch, hz, hws);
        a.hues[k] = hz;
    }
}

struct NeutralConsts { float whitept, rhue, bhue, yhue, yrange, rrange, brange, sc[3]; };
__device__ __forceinline__ NeutralConsts neutral_consts(const NeutralArgs &a)
{
    NeutralConsts k;
    k.whitept = 65535.f * a.whitecoeff;
    k.rhue = a.hues[0]; k.bhue = a.hues[1]; k.yhue = a.hues[2];
    const float ohue = a.hues[3];
    k.yrange = fabsf(ohue - k.yhue) * 0.8f; k.rrange = fabsf(ohue - k.rhue); k.brange = k.rrange;
    const float dl[3] = {1.1f, 1.2f, 1.5f}, th[3] = {0.85f, 0.75f, 0.95f};
#pragma unroll
    for (int i = 0; i < 3; ++i) k.sc[i] = (1.f - th[i]) / sqrtf(dl[i] - 1.f);
    return k;
}
template <bool PC>
__device__ __forceinline__ void neutral_px(const NeutralArgs &a, const NeutralConsts &k, const PqTab pq, size_t o, float r0, float g0, float b0)
{
    const float whitept = k.whitept, rhue = k.rhue, bhue = k.bhue, yhue = k.yhue, yrange = k.yrange, rrange = k.rrange, brange = k.brange;
    const float th[3] = {0.85f, 0.75f, 0.95f};
    const float *sc = k.sc;
    const float PI_180 = (float)(3.14159265358979323846 / 180.0);
    {
        float rgb[3], jch[3], tv[3];
        rgb[0] = std_max(r0 / 65535.f, 0.f);
        rgb[1] = std_max(g0 / 65535.f, 0.f);
        rgb[2] = std_max(b0 / 65535.f, 0.f);
        rgb2jzczhz(pq, rgb[0], rgb[1], rgb[2], jch[0], jch[1], jch[2], a.ws);
        const float ilum = jch[0];
        float hue = jch[2];
        const float iY = (rgb[0] + rgb[1] + rgb[2]) / 3.f;
        mat_vec(a.to_out, rgb, tv); rgb[0] = tv[0]; rgb[1] = tv[1]; rgb[2] = tv[2];
        const float ac = std_max(std_max(rgb[0], rgb[1]), rgb[2]);
        float d[3] = {0.f, 0.f, 0.f};
        const float aac = fabsf(ac);
        if (ac != 0.f) {
            d[0] = (ac - rgb[0]) / aac;
            d[1] = (ac - rgb[1]) / aac;
            d[2] = (ac - rgb[2]) / aac;
        }
        float cd[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            cd[i] = d[i] < th[i] ? d[i] : sc[i] * sqrtf(d[i] - th[i] + sqr(sc[i]) / 4.0f) - sc[i] * sqrtf(sqr(sc[i]) / 4.0f) + th[i];
        rgb[0] = ac - cd[0] * aac;
        rgb[1] = ac - cd[1] * aac;
        rgb[2] = ac - cd[2] * aac;
        mat_vec(a.to_work, rgb, tv); rgb[0] = tv[0]; rgb[1] = tv[1]; rgb[2] = tv[2];
        const float oY = (rgb[0] + rgb[1] + rgb[2]) / 3.f;
        if (oY > 0.f) {
            const float f = iY / oY;
            rgb[0] *= f; rgb[1] *= f; rgb[2] *= f;
            filmlike_clip_dev(rgb[0], rgb[1], rgb[2], whitept);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float nt = rgb[j] * 65535.f;
            nt = (a.tail_kind && nt > 65535.f) ? curve_tail<PC>(a.tail_kind, a.tail_y, a.tail_pc, nt) : lutf_lookup<true>(a.lut, 65536, std_max(nt, 0.f));   // setLutVal
            rgb[j] = nt / 65535.f;
        }
        rgb2jzczhz(pq, rgb[0], rgb[1], rgb[2], jch[0], jch[1], jch[2], a.ws);
        float hue_shift = 15.f * PI_180 * gauss(hue, rhue, rrange);
        hue_shift += -5.f * PI_180 * gauss(hue, bhue, brange);
        hue_shift *= lim01((rgb[0] + rgb[1] + rgb[2]) / (3.f * a.whitecoeff));
        hue += hue_shift;
        float sat = jch[1];
        {
            const float olum = jch[0];
            float ccf = ilum > 1e-5f ? (1.f - (lim01((olum / ilum) - 1.f) * 0.2f)) : 1.f;
            ccf = lim01(ccf + 0.5f * gauss(hue, yhue, yrange));
            sat *= ccf;
        }
        jzczhz2rgb(a.pq_inv, jch[0], sat, hue, rgb[0], rgb[1], rgb[2], a.iws);
        a.img[0][o] = std_max(0.f, std_min(rgb[0] * 65535.f, whitept));
        a.img[1][o] = std_max(0.f, std_min(rgb[1] * 65535.f, whitept));
        a.img[2][o] = std_max(0.f, std_min(rgb[2] * 65535.f, whitept));
    }
}
template <bool PC>
__global__ void __launch_bounds__(256) tone_neutral_kernel(NeutralArgs a)
{
    const NeutralConsts k = neutral_consts(a);
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        const size_t o = (size_t)y * a.stride + x;
        neutral_px<PC>(a, k, PqTab{a.pq, nullptr}, o, a.img[0][o], a.img[1][o], a.img[2][o]);
    }
}
// large frames: one persistent 1024-thread workgroup per CU, the lower 40 704 entries of the forward PQ table in LDS
template <bool PC>
__global__ void __launch_bounds__(1024) tone_neutral_lds_kernel(NeutralArgs a)
{
    extern __shared__ float pq_lds[];
    lut_lds_fill(pq_lds, a.pq, 1024);
    const NeutralConsts k = neutral_consts(a);
    const PqTab pq = {a.pq, pq_lds};
    for (int y = blockIdx.x; y < a.h; y += gridDim.x)
        for (int x0 = 0; x0 < a.w; x0 += 2048) {
            float r[2], g[2], b[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int x = x0 + q * 1024 + (int)threadIdx.x;
                const size_t o = (size_t)y * a.stride + (x < a.w ? x : a.w - 1);
                r[q] = a.img[0][o]; g[q] = a.img[1][o]; b[q] = a.img[2][o];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int x = x0 + q * 1024 + (int)threadIdx.x;
                if (x < a.w) neutral_px<PC>(a, k, pq, (size_t)y * a.stride + x, r[q], g[q], b[q]);
            }
        }
}

hipError_t launch_neutral_hues(const NeutralArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(neutral_hues_kernel, dim3(1), dim3(64), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_tone_neutral(const NeutralArgs &a, hipStream_t s)
{
    if ((long long)a.w * a.h >= (1 << 22) && (reinterpret_cast<uintptr_t>(a.pq) & 15) == 0 && !a.no_lds_lut && device_block_fits(LUT_LDS_N * (int)sizeof(float), 1024)) {
        const size_t lds = (size_t)LUT_LDS_N * sizeof(float);
        const bool pc = a.tail_kind == 4;
        hipError_t e = hipFuncSetAttribute(pc ? reinterpret_cast<const void *>(tone_neutral_lds_kernel<true>) : reinterpret_cast<const void *>(tone_neutral_lds_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (a.cu_reserve > 0) cus = cus - a.cu_reserve > 1 ? cus - a.cu_reserve : 1;
        if (pc) hipLaunchKernelGGL(tone_neutral_lds_kernel<true>, dim3(cus < a.h ? cus : a.h), dim3(1024), lds, s, a);
        else hipLaunchKernelGGL(tone_neutral_lds_kernel<false>, dim3(cus < a.h ? cus : a.h), dim3(1024), lds, s, a);
        return hipGetLastError();
    }
    if (a.tail_kind == 4) hipLaunchKernelGGL(tone_neutral_kernel<true>, image_grid(a.w, a.h), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(tone_neutral_kernel<false>, image_grid(a.w, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}

} // namespace artgpu
