// art_amd/host/rtengine_gpu.h -- C++ host-side mirror of the rtengine classes on the hot path, over the
// C ABI of include/artgpu.h.  Same method names, argument meaning and call order as the reference, so
// that a caller written against rtengine (rtgui/main-cli.cc:786-903, rtengine/simpleprocess.cc:75-420)
// reads the same:
//   RawImageSource::{demosaic, getImage, convertColorSpace, setBorder}   rawimagesource.h:119-136,266-290
//   ImProcFunctions::{process, exposure, toneCurve, denoise}             improcfun.h:95,132-133,147,150
//   Imagefloat (planar fp32 R|G|B, row stride ceil16(W*4))               iimage.h:653-720
// Frames stay resident in HBM between stages (on_device planes).  Error behaviour: the reference
// methods return void; here a failing artgpu_* call throws std::runtime_error with the library's
// message -- an adapter inside ART would instead fall through to the CPU code.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <functional>
#include <vector>
#include "../../include/artgpu.h"

namespace artgpu_host {

inline void hipcheck(hipError_t e, const char *what)
{
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

class Context {
public:
    explicit Context(int device = 0) { if (artgpu_create(device, &ctx_) != 0) throw std::runtime_error("artgpu_create failed"); }
    ~Context() { if (ctx_) artgpu_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    artgpu_ctx *get() const { return ctx_; }
    void check(int rc) const { if (rc != 0) throw std::runtime_error(std::string("artgpu: ") + artgpu_last_error(ctx_)); }
    void synchronize() const { check(artgpu_synchronize(ctx_)); }
private:
    artgpu_ctx *ctx_ = nullptr;
};

// One fp32 plane in HBM (array2D<float> counterpart; rows contiguous, stride W*4)
class DevicePlane {
public:
    DevicePlane() = default;
    DevicePlane(int w, int h, int64_t stride_bytes = 0) { alloc(w, h, stride_bytes); }
    ~DevicePlane() { release(); }
    DevicePlane(const DevicePlane &) = delete;
    DevicePlane &operator=(const DevicePlane &) = delete;
    void alloc(int w, int h, int64_t stride_bytes = 0)
    {
        release();
        w_ = w; h_ = h; stride_ = stride_bytes ? stride_bytes : (int64_t)w * 4;
        hipcheck(hipMalloc(reinterpret_cast<void **>(&p_), (size_t)stride_ * h), "hipMalloc(plane)");
    }
    void release() { if (p_) { (void)hipFree(p_); p_ = nullptr; } }
    void upload(const float *host) { hipcheck(hipMemcpy2D(p_, (size_t)stride_, host, (size_t)w_ * 4, (size_t)w_ * 4, h_, hipMemcpyHostToDevice), "upload"); }
    void download(float *host) const { hipcheck(hipMemcpy2D(host, (size_t)w_ * 4, p_, (size_t)stride_, (size_t)w_ * 4, h_, hipMemcpyDeviceToHost), "download"); }
    artgpu_plane view() const { return artgpu_plane{p_, w_, h_, stride_, 1}; }
    int width() const { return w_; }
    int height() const { return h_; }
private:
    float *p_ = nullptr;
    int w_ = 0, h_ = 0;
    int64_t stride_ = 0;
};

// rtengine::Imagefloat counterpart: three planes, row stride ceil16(W*4) bytes (iimage.h:653-720)
class Imagefloat {
public:
    Imagefloat(int w, int h) : r(w, h, ((int64_t)w * 4 + 15) / 16 * 16), g(w, h, ((int64_t)w * 4 + 15) / 16 * 16), b(w, h, ((int64_t)w * 4 + 15) / 16 * 16) {}
    int getWidth() const { return r.width(); }
    int getHeight() const { return r.height(); }
    artgpu_rgb view() const { return artgpu_rgb{r.view(), g.view(), b.view()}; }
    DevicePlane r, g, b;
};

// The fields of ProcParams the path reads (procparams.cc:1528-3335 for the defaults)
struct ProcParams {
    // raw.bayersensor.{method,border,dualDemosaicContrast,dualDemosaicAutoContrast}; dual = the AMAZEBILINEAR / RCDBILINEAR / AMAZEVNG4 / RCDVNG4 methods
    // (first demosaicer `method`, dualSecond in flat regions; rawimagesource.cc:1876-1886 -> dual_demosaic_RT)
    struct { int method = ARTGPU_BAYER_AMAZE; int border = 4; bool dual = false; int dualSecond = ARTGPU_DUAL_BILINEAR; double dualDemosaicContrast = 20; bool dualDemosaicAutoContrast = true; } bayersensor;
    enum XTransMethod { ONE_PASS = 1, THREE_PASS = 3 };
    struct { int method = THREE_PASS; int border = 7; } xtranssensor;          // raw.xtranssensor.{method,border} (procparams.cc:3064)
    struct { bool enabled = false; double luminance = 0, luminanceDetail = 0; int luminanceDetailThreshold = 0; double chrominance = 15,
             chrominanceRedGreen = 0, chrominanceBlueYellow = 0, gamma = 1.7, chrominanceAutoFactor = 1; bool aggressive = false; int colorSpace = 0, chrominanceMethod = 0;
             bool smoothingEnabled = false; int guidedChromaRadius = 3, nlDetail = 80, nlStrength = 0; } denoise;   // procparams.cc:1900-1918 (chrominanceMethod: 0 MANUAL, 1 AUTOMATIC)
    struct { bool enabled = true; double expcomp = 0, black = 0; } exposure;
    struct { bool enabled = false; int red[3] = {1000, 0, 0}, green[3] = {0, 1000, 0}, blue[3] = {0, 0, 1000}; } chmixer;        // procparams.cc (ChannelMixerParams)
    struct { bool enabled = false; std::vector<double> hCurve, sCurve, lCurve; int smoothing = 0; } hsl;                         // HSLEqualizerParams
    struct { bool enabled = false; double gain = 0, targetGray = 18, blackEv = -13.5, whiteEv = 2.5; int regularization = 60; bool satcontrol = true;
             int highlightCompression = 0; } logenc;                                                                               // LogEncodingParams (procparams.cc:2039-2051)
    // LabCurveParams: brightness / contrast / the three DiagonalCurves stay host code (get_L_curve / get_ab_curves, iplabadjustments.cc:67-191);
    // `curves` is that code: it receives hist16 (nullptr unless contrast != 0) and fills lcurve[32770], acurve[65536], bcurve[65536]
    struct { bool enabled = false; int chromaticity = 0, contrast = 0;
             std::function<void(const uint32_t *hist16, std::vector<float> &lcurve, std::vector<float> &acurve, std::vector<float> &bcurve)> curves; } labCurve;
    struct { bool enabled = false; int saturation = 0, vibrance = 0; } saturation;                                               // SaturationParams
    struct { bool enabled = false; std::vector<float> rlut, glut, blut; } rgbCurves;                                             // RGBCurvesParams, as outCurve LUTs
    struct { bool enabled = true; int curveMode = ARTGPU_TONE_STD; std::vector<float> lut; float whitePoint = 1.f; bool basecurveLinear = true; } toneCurve;
    // toneCurve.curveMode: ARTGPU_TONE_STD or ARTGPU_TONE_NEUTRAL (ART's default, procparams.cc:1585)
    double workingSpace[9] = {0.6734241, 0.1656411, 0.1251286, 0.2790177, 0.6753402, 0.0456377, -0.0019300, 0.0299784, 0.7973330}; // Rec2020 TMatrix (iccmatrices.h:151-155)
    double workingSpaceInverse[9] = {1.6473376, -0.3935675, -0.2359961, -0.6826036, 1.6475887, 0.0128190, 0.0296524, -0.0628993, 1.2531279};   // iccmatrices.h:157-161
    // NeutralToneCurve::ApplyState::to_out / to_work (curves.cc:870-878): identity unless the output profile has a matrix
    float toOut[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, toWork[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
};

// rtengine::RawImageSource counterpart for a Bayer sensor
class RawImageSource {
public:
    RawImageSource(Context &c, int W, int H, uint32_t filters, double initialGain = 1.0)
        : ctx(c), W(W), H(H), filters(filters), initialGain(initialGain), rawData(W, H), red(W, H), green(W, H), blue(W, H) {}
    // X-Trans sensor: 6x6 colour map (RawImage::getXtransMatrix) and camera matrix (getRgbCam)
    RawImageSource(Context &c, int W, int H, const int32_t xtrans_[36], const float rgb_cam_[12])
        : ctx(c), W(W), H(H), filters(9), initialGain(1.0), rawData(W, H), red(W, H), green(W, H), blue(W, H), isXtrans(true)
    {
        for (int i = 0; i < 36; ++i) xtrans[i] = xtrans_[i];
        for (int i = 0; i < 12; ++i) rgb_cam[i] = rgb_cam_[i];
        border = 7;
    }
    void setBorder(int b) { border = b; }                                   // rawimagesource.h (simpleprocess.cc:138-146)
    void load(const float *cfa_host) { rawData.upload(cfa_host); }          // stands in for load()/preprocess(): CFA 0..65535
    // RawImageSource::demosaic (rawimagesource.cc:1854-1962)
    void demosaic(const ProcParams &p)
    {
        artgpu_plane raw = rawData.view();
        artgpu_rgb out{red.view(), green.view(), blue.view()};
        if (isXtrans) {   // rawimagesource.cc:1915-1925: ONE_PASS -> (1, false), THREE_PASS -> (3, true)
            const int passes = p.xtranssensor.method == ProcParams::ONE_PASS ? 1 : 3;
            ctx.check(artgpu_demosaic_xtrans(ctx.get(), passes, passes > 1 ? 1 : 0, &raw, xtrans, rgb_cam, &out));
            return;
        }
        if (p.bayersensor.dual) {       // `contrast` comes back as the threshold in use (the reference's `double &contrast`)
            dualDemosaicContrastUsed = p.bayersensor.dualDemosaicContrast;
            ctx.check(artgpu_dual_demosaic_bayer(ctx.get(), p.bayersensor.method, p.bayersensor.dualSecond, &raw, filters, initialGain, border, &dualDemosaicContrastUsed,
                                                 p.bayersensor.dualDemosaicAutoContrast ? 1 : 0, &out));
            return;
        }
        ctx.check(artgpu_demosaic_bayer(ctx.get(), p.bayersensor.method, &raw, filters, initialGain, border, &out));
    }
    double dualDemosaicContrastUsed = 0;
    void getFullSize(int &w, int &h) const { w = W - 2 * border; h = H - 2 * border; }   // computeFullSize (L1163-1193), tran = 0
    // RawImageSource::getImage (rawimagesource.cc:781-1104): tran = 0, skip = 1; rm/gm/bm as the caller computed them
    void getImage(const float mul[3], bool doClip, Imagefloat *image, int x = 0, int y = 0, int skip = 1)
    {
        artgpu_rgb planes{red.view(), green.view(), blue.view()};
        artgpu_rgb img = image->view();
        // PreviewProps(x, y, w, h, skip): transformRect adds the border; rm/gm/bm are divided by skip*skip (L922-926)
        const float area = float(skip) * float(skip);
        const float m[3] = {mul[0] / area, mul[1] / area, mul[2] / area};
        ctx.check(artgpu_get_image_skip(ctx.get(), &planes, x + border, y + border, skip, skip > 1 ? m : mul, doClip ? 1 : 0, nullptr, &img));
    }
    // RawImageSource::convertColorSpace, matrix branch (rawimagesource.cc:1128-1143,3184-3213)
    void convertColorSpace(Imagefloat *image, const double mat[9])
    {
        setColorMatrix(mat);
        artgpu_rgb img = image->view();
        ctx.check(artgpu_convert_color_space(ctx.get(), &img, mat));
    }
    // the camera -> working-space matrix convertColorSpace derives from params->icm (rawimagesource.cc:1128-1143); remembered
    // because ImProcFunctions::denoise applies it again to its quarter-size calclum copy (ipdenoise.cc:1131)
    void setColorMatrix(const double mat[9]) { for (int k = 0; k < 9; ++k) colorMatrix[k] = mat[k]; hasColorMatrix = true; }
    double colorMatrix[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    bool hasColorMatrix = false;
    Context &ctx;
    int W, H, border = 4;
    uint32_t filters;
    double initialGain;
    DevicePlane rawData, red, green, blue;
    bool isXtrans = false;
    int32_t xtrans[36] = {};
    float rgb_cam[12] = {};
};

// rtengine::ImProcFunctions counterpart
class ImProcFunctions {
public:
    enum class Stage { STAGE_0, STAGE_1, STAGE_2, STAGE_3 };
    enum class Pipeline { THUMBNAIL, NAVIGATOR, PREVIEW, OUTPUT };
    ImProcFunctions(Context &c, const ProcParams *p, double scale = 1.0) : ctx(c), params(p), scale(scale) {}

    // ImProcFunctions::process (improcfun.cc:567-641): same step order; steps that are disabled / identity in the
    // default ProcParams (dehaze, DRC, channelMixer, hslEqualizer, toneEqualizer, sharpening, ... blackAndWhite) are
    // not on the device path and are skipped here exactly as their `enabled == false` early-outs skip them.
    bool process(Pipeline, Stage stage, Imagefloat *img)
    {
        switch (stage) {
        case Stage::STAGE_0: break;
        case Stage::STAGE_1: channelMixer(img); exposure(img); hslEqualizer(img); break;      // improcfun.cc:581-585
        case Stage::STAGE_2: break;
        case Stage::STAGE_3: logEncoding(img); saturationVibrance(img); toneCurve(img); rgbCurves(img); labAdjustments(img); break;  // improcfun.cc:607-623 (the steps this library has)
        }
        return false;
    }
    // ImProcFunctions::channelMixer (ipchmixer.cc:152-234), RGB_MATRIX mode: the nine percentages / 1000 (L185-199)
    void channelMixer(Imagefloat *img)
    {
        if (!params->chmixer.enabled) return;
        float m[9];
        for (int k = 0; k < 3; ++k) { m[k] = params->chmixer.red[k] / 1000.f; m[3 + k] = params->chmixer.green[k] / 1000.f; m[6 + k] = params->chmixer.blue[k] / 1000.f; }
        artgpu_rgb i = img->view();
        ctx.check(artgpu_channel_mixer(ctx.get(), &i, m));
    }
    // ImProcFunctions::hslEqualizer (iphsl.cc:29-221); leaves the image in YUV mode like the reference unless the caller's Imagefloat
    // has no mode tracking (this mirror's has none: planes are converted back to RGB)
    void hslEqualizer(Imagefloat *img)
    {
        const auto &p = params->hsl;
        if (!p.enabled) return;
        artgpu_rgb i = img->view();
        ctx.check(artgpu_hsl_equalizer(ctx.get(), &i, p.hCurve.empty() ? nullptr : p.hCurve.data(), (int)p.hCurve.size(), p.sCurve.empty() ? nullptr : p.sCurve.data(),
                                       (int)p.sCurve.size(), p.lCurve.empty() ? nullptr : p.lCurve.data(), (int)p.lCurve.size(), p.smoothing, params->workingSpace, scale, 1));
    }
    // ImProcFunctions::logEncoding (iplogenc.cc:395-402); full_width/full_height as set by ImProcFunctions::setViewport (0 = the image itself)
    int full_width = 0, full_height = 0;
    void logEncoding(Imagefloat *img)
    {
        const auto &p = params->logenc;
        if (!p.enabled) return;
        artgpu_logenc_params lp = {1, p.regularization, p.satcontrol ? 1 : 0, p.highlightCompression, p.gain, p.targetGray, p.blackEv, p.whiteEv};
        artgpu_rgb i = img->view();
        ctx.check(artgpu_log_encoding(ctx.get(), &i, &lp, params->workingSpace, full_width, full_height));
    }
    // ImProcFunctions::labAdjustments (iplabadjustments.cc:277-345): setMode(LAB), hist16 when contrast != 0, the caller's curve builders,
    // the curve loop; the reference leaves the image in LAB mode for the next step's setMode -- this mirror has no mode tracking and
    // converts back at once (Imagefloat::lab_to_rgb)
    void labAdjustments(Imagefloat *img)
    {
        const auto &p = params->labCurve;
        if (!p.enabled || !p.curves) return;
        artgpu_rgb i = img->view();
        ctx.check(artgpu_rgb_to_lab(ctx.get(), &i, params->workingSpace));
        std::vector<uint32_t> hist16;
        if (p.contrast != 0) { hist16.resize(65536); ctx.check(artgpu_lab_histogram(ctx.get(), &i, hist16.data())); }
        std::vector<float> lc, ac, bc;
        p.curves(hist16.empty() ? nullptr : hist16.data(), lc, ac, bc);
        if (lc.size() != 32770 || ac.size() != 65536 || bc.size() != 65536) throw std::runtime_error("labAdjustments: curve LUT sizes");
        ctx.check(artgpu_lab_adjustments(ctx.get(), &i, lc.data(), ac.data(), bc.data(), (p.chromaticity + 100.0f) / 100.0f));
        ctx.check(artgpu_lab_to_rgb(ctx.get(), &i, params->workingSpaceInverse));
    }
    // ImProcFunctions::saturationVibrance (ipsaturation.cc:43-83)
    void saturationVibrance(Imagefloat *img)
    {
        if (!params->saturation.enabled) return;
        artgpu_rgb i = img->view();
        ctx.check(artgpu_saturation_vibrance(ctx.get(), &i, params->saturation.saturation, params->saturation.vibrance, params->workingSpace));
    }
    // ImProcFunctions::rgbCurves (iprgbcurves.cc:30-148): the three 65536-entry outCurve LUTs are built by the caller (RGBCurve, L41-53)
    void rgbCurves(Imagefloat *img)
    {
        const auto &p = params->rgbCurves;
        if (!p.enabled) return;
        artgpu_rgb i = img->view();
        ctx.check(artgpu_rgb_curves(ctx.get(), &i, p.rlut.size() == 65536 ? p.rlut.data() : nullptr, p.glut.size() == 65536 ? p.glut.data() : nullptr,
                                    p.blut.size() == 65536 ? p.blut.data() : nullptr));
    }
    // ImProcFunctions::exposure -> expcomp (ipexposure.cc:28-79)
    void exposure(Imagefloat *img) { expcomp(img, params->exposure.expcomp, params->exposure.black, params->exposure.enabled); }
    void expcomp(Imagefloat *img, double ec, double black, bool enabled)
    {
        if (!enabled) return;
        const float exp_scale = (float)std::pow(2.0, ec);          // pow(2.f, double) promotes (ipexposure.cc:40)
        const float blk = (float)(black * 2000.0);
        artgpu_rgb i = img->view();
        ctx.check(artgpu_exposure(ctx.get(), &i, exp_scale, blk));
    }
    // ImProcFunctions::toneCurve (iptonecurve.cc:553-716): single STD curve, LINEAR base curve
    void toneCurve(Imagefloat *img)
    {
        if (!params->toneCurve.enabled) return;
        artgpu_rgb i = img->view();
        const float *lut = params->toneCurve.lut.size() == 65536 ? params->toneCurve.lut.data() : nullptr;
        if (params->toneCurve.curveMode == ARTGPU_TONE_NEUTRAL) {
            // single NEUTRAL curve: no filmlike_clip pre-pass (iptonecurve.cc:586-595), apply_tc(NEUTRAL) with basecurve == nullptr
            artgpu_neutral_state st;
            for (int k = 0; k < 9; ++k) { st.ws[k] = params->workingSpace[k]; st.iws[k] = params->workingSpaceInverse[k]; st.to_out[k] = params->toOut[k]; st.to_work[k] = params->toWork[k]; }
            ctx.check(artgpu_tone_curve_neutral(ctx.get(), &i, lut, params->toneCurve.whitePoint, &st));
            return;
        }
        ctx.check(artgpu_tone_curve(ctx.get(), &i, params->toneCurve.curveMode, lut, params->toneCurve.whitePoint, params->toneCurve.basecurveLinear ? 1 : 0));
    }
    // DenoiseInfoStore (improcfun.h:117-131)
    struct DenoiseInfoStore : artgpu_denoise_info_store {
        DenoiseInfoStore() { reset(); }
        void reset() { *static_cast<artgpu_denoise_info_store *>(this) = artgpu_denoise_info_store{}; }
    };
    // ImProcFunctions::denoiseComputeParams (ipdenoise.cc:800-1093): AUTOMATIC chrominance from nine crops of the sensor planes.
    // The reference pulls rm/gm/bm and the camera matrix out of imgsrc/currWB/params->icm; here the caller hands in the
    // multipliers and must have called imgsrc->setColorMatrix().
    void denoiseComputeParams(RawImageSource *imgsrc, const float mul[3], bool doClip, DenoiseInfoStore &store, decltype(ProcParams::denoise) &dnparams)
    {
        artgpu_denoise_params dn{dnparams.luminance, dnparams.luminanceDetail, dnparams.luminanceDetailThreshold, dnparams.chrominance,
                                 dnparams.chrominanceRedGreen, dnparams.chrominanceBlueYellow, dnparams.gamma, dnparams.aggressive ? 1 : 0,
                                 dnparams.colorSpace, dnparams.chrominanceMethod};
        artgpu_rgb planes{imgsrc->red.view(), imgsrc->green.view(), imgsrc->blue.view()};
        ctx.check(artgpu_denoise_compute_params(ctx.get(), &planes, imgsrc->border, mul, doClip ? 1 : 0, imgsrc->colorMatrix, params->workingSpace,
                                                dnparams.chrominanceAutoFactor, &store, &dn));
        dnparams.chrominance = dn.chrominance; dnparams.chrominanceRedGreen = dn.chrominance_red_green; dnparams.chrominanceBlueYellow = dn.chrominance_blue_yellow;
    }
    // ImProcFunctions::denoise (ipdenoise.cc:1096-1189): calclum/ccalc chroma noise map with the fixed noise curve
    // (L1139-1149), exposure pre-compensation, RGB_denoise, guided smoothing + NL-means, post-compensation.
    void denoise(RawImageSource *imgsrc, Imagefloat *img)
    {
        const auto &d = params->denoise;
        if (!d.enabled) return;
        const double ecomp = params->exposure.enabled ? params->exposure.expcomp : 0.0;
        artgpu_denoise_tool_params tp{{d.luminance, d.luminanceDetail, d.luminanceDetailThreshold, d.chrominance, d.chrominanceRedGreen,
                                       d.chrominanceBlueYellow, d.gamma, d.aggressive ? 1 : 0, d.colorSpace, d.chrominanceMethod},
                                      d.smoothingEnabled ? 1 : 0, d.guidedChromaRadius, d.nlStrength, d.nlDetail};
        static const double curve_points[9] = {1 /*FCT_MinMaxCPoints*/, 0.05, 0.50, 0.35, 0.35, 0.35, 0.05, 0.35, 0.35};
        float curve[501], sum = 0.f;
        ctx.check(artgpu_noise_curve_lut(curve_points, 9, curve, &sum));
        artgpu_rgb i = img->view();
        ctx.check(artgpu_improc_denoise(ctx.get(), &i, &tp, params->workingSpace, params->workingSpaceInverse, ecomp, scale, imgsrc && imgsrc->hasColorMatrix ? imgsrc->colorMatrix : nullptr,
                                        curve, 0u));
    }
    Context &ctx;
    const ProcParams *params;
    double scale;
};

// The batch queue's loop (rtengine/simpleprocess.cc:586-612 batchProcessingThread: one ImageProcessor per job, jobs share nothing) between
// the decoder's and the writers' formats: uint16 sensor data in, the scanlines Imagefloat::getScanline hands the TIFF / PNG writers out
// (artgpu_batch_run_io).  Buffers are pinned (hipHostMalloc) so that a job's upload and download run beside its neighbours' kernels.
class BatchQueue {
public:
    struct Job {
        uint16_t *sensor = nullptr;      // W x H, pinned
        unsigned char *scanlines = nullptr;   // (H - 2 border) x (W - 2 border) x 3 x bps / 8, pinned
        int W = 0, H = 0;
        float chmax[4] = {0, 0, 0, 0};
        int status = 0;
    };
    BatchQueue(Context &c, int bps = 16) : ctx(c), bps(bps) {}
    ~BatchQueue() { for (Job &j : jobs) { if (j.sensor) (void)hipHostFree(j.sensor); if (j.scanlines) (void)hipHostFree(j.scanlines); } }
    BatchQueue(const BatchQueue &) = delete;
    BatchQueue &operator=(const BatchQueue &) = delete;
    // a job's buffers; the caller fills `sensor` (the decoder's output)
    Job &addJob(int W, int H, int border)
    {
        Job j; j.W = W; j.H = H;
        hipcheck(hipHostMalloc(reinterpret_cast<void **>(&j.sensor), (size_t)W * H * 2, hipHostMallocDefault), "hipHostMalloc(sensor)");
        hipcheck(hipHostMalloc(reinterpret_cast<void **>(&j.scanlines), rowBytes(W, border) * (size_t)(H - 2 * border), hipHostMallocDefault), "hipHostMalloc(scanlines)");
        jobs.push_back(j);
        return jobs.back();
    }
    size_t rowBytes(int W, int border) const { return (size_t)(W - 2 * border) * 3 * (bps / 8); }
    // ImageProcessor::operator() for every job: stage_init (scaleColors + demosaic), stage_denoise, stage_finish up to the tone curve, then the
    // output profile's matrix (identity here: the working space is the output space) and getScanline
    void process(const ProcParams &p, uint32_t filters, const float mul[3], const double cam2work[9], int lanes,
                 const float cblacksom[4], const float scale_mul[4])
    {
        artgpu_pipeline_params pp = {};
        pp.sensor = 0; pp.bayer_method = p.bayersensor.method; pp.filters = filters; pp.initial_gain = 1.0; pp.xtrans_passes = 3; pp.border = p.bayersensor.border;
        for (int k = 0; k < 3; ++k) pp.mul[k] = mul[k];
        pp.do_clip = 1; pp.has_cam_to_work = 1;
        for (int k = 0; k < 9; ++k) { pp.cam_to_work[k] = cam2work[k]; pp.ws[k] = p.workingSpace[k]; pp.iws[k] = p.workingSpaceInverse[k]; pp.to_out[k] = p.toOut[k]; pp.to_work[k] = p.toWork[k]; }
        const auto &d = p.denoise;
        pp.denoise_enabled = d.enabled ? 1 : 0;
        pp.denoise = artgpu_denoise_tool_params{{d.luminance, d.luminanceDetail, d.luminanceDetailThreshold, d.chrominance, d.chrominanceRedGreen, d.chrominanceBlueYellow,
                                                 d.gamma, d.aggressive ? 1 : 0, d.colorSpace, d.chrominanceMethod},
                                                d.smoothingEnabled ? 1 : 0, d.guidedChromaRadius, d.nlStrength, d.nlDetail};
        pp.exposure_enabled = p.exposure.enabled ? 1 : 0; pp.expcomp = p.exposure.expcomp; pp.black = p.exposure.black;
        pp.tone_enabled = p.toneCurve.enabled ? 1 : 0; pp.tone_mode = p.toneCurve.curveMode; pp.tone_lut = p.toneCurve.lut.data(); pp.white_point = p.toneCurve.whitePoint;
        pp.scale = 1.0; pp.chrominance_auto_factor = d.chrominanceAutoFactor;
        std::vector<artgpu_sensor_frame> in(jobs.size());
        std::vector<artgpu_scanline_frame> out(jobs.size());
        for (size_t k = 0; k < jobs.size(); ++k) {
            const Job &j = jobs[k];
            in[k] = artgpu_sensor_frame{j.sensor, j.W, j.H, (int64_t)j.W * 2, 1, 0, {cblacksom[0], cblacksom[1], cblacksom[2], cblacksom[3]},
                                        {scale_mul[0], scale_mul[1], scale_mul[2], scale_mul[3]}};
            out[k] = artgpu_scanline_frame{};
            out[k].scanlines = j.scanlines; out[k].row_stride_bytes = (int64_t)rowBytes(j.W, pp.border); out[k].bps = bps; out[k].is_float = 0;
            out[k].rgb2out_enabled = 0; out[k].trc_linear = 1;
        }
        ctx.check(artgpu_set_batch_lanes(ctx.get(), lanes));
        const int rc = artgpu_batch_run_io(ctx.get(), (int)jobs.size(), in.data(), &pp, out.data());
        for (size_t k = 0; k < jobs.size(); ++k) { jobs[k].status = out[k].status; for (int c = 0; c < 4; ++c) jobs[k].chmax[c] = out[k].chmax[c]; }
        ctx.check(rc);
    }
    std::vector<Job> jobs;
    Context &ctx;
    int bps;
};

} // namespace artgpu_host
