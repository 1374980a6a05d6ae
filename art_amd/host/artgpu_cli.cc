// art_amd/host/artgpu_cli.cc -- command-line counterpart of ART-cli for the hot path
// (reference: rtgui/main-cli.cc:786-903 -> rtengine/simpleprocess.cc:75-420): load a CFA frame,
// run stage_init (demosaic, getImage), stage_denoise (convertColorSpace, denoise), stage_finish
// (process STAGE_1..3) on the GPU through the rtengine-shaped classes of rtengine_gpu.h, write a
// 16-bit PPM.  Input: raw little-endian float32 (values 0..65535) or uint16, W*H samples.
//
//   artgpu-cli --in frame.f32 --width 4000 --height 3000 [--u16] [--filters 0x94949494]
//              [--method amaze|rcd] [--border 4] [--denoise L,C] [--chroma-auto] [--expcomp 0.3] [--out out.ppm]
//              [--dual bilinear|vng4] [--dual-contrast C] [--logenc REG] [--saturation S,V] [--labchroma C]   (SURVEY 8f N4 tools)
//   artgpu-cli --batch a.u16,b.u16,... --width W --height H [--lanes N] [--black B] [--method ..] [--denoise L,C] [--expcomp E] [--out prefix]
//              the batch queue's loop (simpleprocess.cc:586-612): uint16 sensor frames through scaleColors + the same stages, 16-bit
//              scanlines as the writers take them (getScanline: clip and truncate), written as prefix.K.ppm; artgpu_batch_run_io
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>
#include "rtengine_gpu.h"

using namespace artgpu_host;

static std::vector<float> default_tone_lut()
{
    // a fixed S-curve (an input of the tone stage; ART builds it from the user's DiagonalCurve, curves.cc:221-231)
    std::vector<float> lut(65536);
    for (int i = 0; i < 65536; ++i) {
        const double x = i / 65535.0;
        lut[i] = (float)((1.0 - std::cos(M_PI * std::pow(x, 0.7))) / 2.0 * 65535.0);
    }
    return lut;
}

int main(int argc, char **argv)
{
    std::string in, out, batch;
    int lanes = 1;
    float black = 0.f;
    int W = 0, H = 0, border = 4, method = ARTGPU_BAYER_AMAZE;
    bool u16 = false;
    uint32_t filters = 0x94949494u;
    double lum = 0, chroma = 0, expcomp = 0;
    bool dn = false, smoothing = false, chroma_auto = false;
    int tone_mode = ARTGPU_TONE_STD;
    int xtrans_passes = 0;   // 0 = Bayer; 1 / 3 = X-Trans ONE_PASS / THREE_PASS with the Fuji colour map
    int gradius = 3, nlstrength = 0, nldetail = 80;
    bool dual = false, dual_auto = true, logenc = false, labcurve = false;
    int dual_second = ARTGPU_DUAL_BILINEAR, logenc_reg = 60, sat = 0, vib = 0, labchroma = 0;
    double dual_contrast = 20;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--in") in = next();
        else if (a == "--batch") batch = next();
        else if (a == "--lanes") lanes = std::atoi(next());
        else if (a == "--black") black = (float)std::atof(next());
        else if (a == "--out") out = next();
        else if (a == "--width") W = std::atoi(next());
        else if (a == "--height") H = std::atoi(next());
        else if (a == "--u16") u16 = true;
        else if (a == "--filters") filters = (uint32_t)std::strtoul(next(), nullptr, 0);
        else if (a == "--border") border = std::atoi(next());
        else if (a == "--expcomp") expcomp = std::atof(next());
        else if (a == "--method") { std::string m = next(); method = (m == "rcd") ? ARTGPU_BAYER_RCD : ARTGPU_BAYER_AMAZE; }
        else if (a == "--chroma-auto") { chroma_auto = true; }
        else if (a == "--denoise") { dn = true; if (std::sscanf(next(), "%lf,%lf", &lum, &chroma) != 2) { std::fprintf(stderr, "--denoise L,C\n"); return 2; } }
        else if (a == "--xtrans") { xtrans_passes = std::atoi(next()); border = 7; }
        else if (a == "--tone") { std::string m = next(); tone_mode = (m == "neutral") ? ARTGPU_TONE_NEUTRAL : ARTGPU_TONE_STD; }
        else if (a == "--smoothing") { smoothing = true; if (std::sscanf(next(), "%d,%d,%d", &gradius, &nlstrength, &nldetail) != 3) { std::fprintf(stderr, "--smoothing radius,nlStrength,nlDetail\n"); return 2; } }
        else if (a == "--dual") { std::string m = next(); dual = true; dual_second = (m == "vng4") ? ARTGPU_DUAL_VNG4 : ARTGPU_DUAL_BILINEAR; }
        else if (a == "--dual-contrast") { dual_contrast = std::atof(next()); dual_auto = false; }
        else if (a == "--logenc") { logenc = true; logenc_reg = std::atoi(next()); }
        else if (a == "--saturation") { if (std::sscanf(next(), "%d,%d", &sat, &vib) != 2) { std::fprintf(stderr, "--saturation S,V\n"); return 2; } }
        else if (a == "--labchroma") { labchroma = std::atoi(next()); labcurve = true; }
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if ((in.empty() && batch.empty()) || W <= 0 || H <= 0) { std::fprintf(stderr, "usage: artgpu-cli --in frame.f32 | --batch a.u16,b.u16,... --width W --height H [options]\n"); return 2; }
    if (!batch.empty()) {
        try {
            Context ctx(0);
            ProcParams params;
            params.bayersensor.method = method; params.bayersensor.border = border;
            params.denoise.enabled = dn; params.denoise.luminance = lum; params.denoise.chrominance = chroma; params.denoise.luminanceDetail = 50;
            params.exposure.expcomp = expcomp;
            params.toneCurve.lut = default_tone_lut(); params.toneCurve.curveMode = tone_mode;
            BatchQueue q(ctx, 16);
            std::vector<std::string> names;
            for (size_t pos = 0; pos <= batch.size();) {
                const size_t e = batch.find(',', pos);
                names.push_back(batch.substr(pos, e == std::string::npos ? std::string::npos : e - pos));
                if (e == std::string::npos) break;
                pos = e + 1;
            }
            for (const std::string &n : names) {
                BatchQueue::Job &j = q.addJob(W, H, border);
                std::ifstream f(n, std::ios::binary);
                if (!f) throw std::runtime_error("cannot open " + n);
                f.read(reinterpret_cast<char *>(j.sensor), (std::streamsize)W * H * 2);
                if (!f) throw std::runtime_error("short read on " + n);
            }
            const float mul[3] = {2.1374f, 1.0f, 1.5918f};
            const double mat[9] = {0.6325, 0.2312, 0.0921, 0.2198, 0.7712, 0.0090, 0.0166, 0.0713, 0.7514};
            const float cblack[4] = {black, black, black, black}, smul[4] = {1.f, 1.f, 1.f, 1.f};
            auto t0 = std::chrono::steady_clock::now();
            q.process(params, filters, mul, mat, lanes, cblack, smul);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            const int fw = W - 2 * border, fh = H - 2 * border;
            for (size_t k = 0; k < q.jobs.size(); ++k) {
                if (out.empty()) break;
                std::ofstream o(out + "." + std::to_string(k) + ".ppm", std::ios::binary);
                o << "P6\n" << fw << " " << fh << "\n65535\n";
                const uint16_t *sc = reinterpret_cast<const uint16_t *>(q.jobs[k].scanlines);
                std::vector<unsigned char> row((size_t)fw * 6);
                for (int y = 0; y < fh; ++y) {
                    for (size_t e = 0; e < (size_t)fw * 3; ++e) { const uint16_t v = sc[(size_t)y * fw * 3 + e]; row[2 * e] = (unsigned char)(v >> 8); row[2 * e + 1] = (unsigned char)(v & 255); }
                    o.write(reinterpret_cast<char *>(row.data()), (std::streamsize)row.size());
                }
            }
            std::printf("{\"frames\": %zu, \"lanes\": %d, \"width\": %d, \"height\": %d, \"out_width\": %d, \"out_height\": %d, \"batch_ms\": %.3f, \"chmax0\": [%.1f, %.1f, %.1f]}\n",
                        q.jobs.size(), lanes, W, H, fw, fh, ms, q.jobs[0].chmax[0], q.jobs[0].chmax[1], q.jobs[0].chmax[2]);
        } catch (const std::exception &e) {
            std::fprintf(stderr, "artgpu-cli: %s\n", e.what());
            return 1;
        }
        return 0;
    }
    try {
        std::vector<float> cfa((size_t)W * H);
        std::ifstream f(in, std::ios::binary);
        if (!f) throw std::runtime_error("cannot open " + in);
        if (u16) {
            std::vector<uint16_t> t((size_t)W * H);
            f.read(reinterpret_cast<char *>(t.data()), (std::streamsize)t.size() * 2);
            for (size_t k = 0; k < t.size(); ++k) cfa[k] = (float)t[k];
        } else {
            f.read(reinterpret_cast<char *>(cfa.data()), (std::streamsize)cfa.size() * 4);
        }
        if (!f) throw std::runtime_error("short read on " + in);

        Context ctx(0);
        ProcParams params;
        params.bayersensor.method = method;
        params.bayersensor.border = border;
        params.denoise.enabled = dn; params.denoise.luminance = lum; params.denoise.chrominance = chroma; params.denoise.luminanceDetail = 50;
        params.denoise.smoothingEnabled = smoothing; params.denoise.guidedChromaRadius = gradius; params.denoise.nlStrength = nlstrength; params.denoise.nlDetail = nldetail;
        params.exposure.expcomp = expcomp;
        params.toneCurve.lut = default_tone_lut();
        params.toneCurve.curveMode = tone_mode;
        // the default-off tools of SURVEY 8f N4, driven through the same ImProcFunctions mirror
        params.bayersensor.dual = dual; params.bayersensor.dualSecond = dual_second;
        params.bayersensor.dualDemosaicContrast = dual_contrast; params.bayersensor.dualDemosaicAutoContrast = dual_auto;
        params.logenc.enabled = logenc; params.logenc.regularization = logenc_reg;
        params.saturation.enabled = sat != 0 || vib != 0; params.saturation.saturation = sat; params.saturation.vibrance = vib;
        params.labCurve.enabled = labcurve; params.labCurve.chromaticity = labchroma;
        params.labCurve.curves = [](const uint32_t *, std::vector<float> &lc, std::vector<float> &ac, std::vector<float> &bc) {
            // stands in for get_L_curve / get_ab_curves (DiagonalCurve, host code of the application): identity curves, so only chromaticity acts
            lc.resize(32770); ac.resize(65536); bc.resize(65536);
            for (int i = 0; i < 32770; ++i) lc[i] = (float)i;
            for (int i = 0; i < 65536; ++i) ac[i] = bc[i] = (float)i;
        };

        auto t0 = std::chrono::steady_clock::now();
        // stage_init (simpleprocess.cc:215-259)
        static const int32_t fuji[36] = {1, 1, 0, 1, 1, 2, 1, 1, 2, 1, 1, 0, 2, 0, 1, 0, 2, 1, 1, 1, 2, 1, 1, 0, 1, 1, 0, 1, 1, 2, 0, 2, 1, 2, 0, 1};
        static const float cam[12] = {1.60f, -0.45f, -0.15f, 0.f, -0.20f, 1.45f, -0.25f, 0.f, 0.02f, -0.50f, 1.48f, 0.f};
        RawImageSource imgsrc = xtrans_passes ? RawImageSource(ctx, W, H, fuji, cam) : RawImageSource(ctx, W, H, filters, 1.0);
        params.xtranssensor.method = xtrans_passes == 1 ? ProcParams::ONE_PASS : ProcParams::THREE_PASS;
        imgsrc.setBorder(border);
        imgsrc.load(cfa.data());
        imgsrc.demosaic(params);
        int fw, fh;
        imgsrc.getFullSize(fw, fh);
        Imagefloat img(fw, fh);
        const float mul[3] = {2.1374f, 1.0f, 1.5918f};
        const double mat[9] = {0.6325, 0.2312, 0.0921, 0.2198, 0.7712, 0.0090, 0.0166, 0.0713, 0.7514};
        ImProcFunctions ipf(ctx, &params, 1.0);
        if (dn && chroma_auto) {    // simpleprocess.cc:254-256
            ImProcFunctions::DenoiseInfoStore dnstore;
            params.denoise.chrominanceMethod = 1;
            imgsrc.setColorMatrix(mat);
            ipf.denoiseComputeParams(&imgsrc, mul, true, dnstore, params.denoise);
            std::fprintf(stderr, "auto chrominance: %.6f  red-green %.6f  blue-yellow %.6f\n", params.denoise.chrominance,
                         params.denoise.chrominanceRedGreen, params.denoise.chrominanceBlueYellow);
        }
        imgsrc.getImage(mul, true, &img);
        // stage_denoise (simpleprocess.cc:311-315)
        imgsrc.convertColorSpace(&img, mat);
        ipf.denoise(&imgsrc, &img);
        // stage_finish (simpleprocess.cc:389-396)
        ipf.process(ImProcFunctions::Pipeline::OUTPUT, ImProcFunctions::Stage::STAGE_1, &img);
        ipf.process(ImProcFunctions::Pipeline::OUTPUT, ImProcFunctions::Stage::STAGE_2, &img);
        ipf.process(ImProcFunctions::Pipeline::OUTPUT, ImProcFunctions::Stage::STAGE_3, &img);
        ctx.synchronize();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

        std::vector<float> r((size_t)fw * fh), g(r.size()), b(r.size());
        img.r.download(r.data()); img.g.download(g.data()); img.b.download(b.data());
        double sum = 0;
        for (size_t k = 0; k < r.size(); ++k) sum += r[k] + g[k] + b[k];
        if (!out.empty()) {
            std::ofstream o(out, std::ios::binary);
            o << "P6\n" << fw << " " << fh << "\n65535\n";
            std::vector<unsigned char> row((size_t)fw * 6);
            for (int y = 0; y < fh; ++y) {
                for (int x = 0; x < fw; ++x) {
                    const float v[3] = {r[(size_t)y * fw + x], g[(size_t)y * fw + x], b[(size_t)y * fw + x]};
                    for (int c = 0; c < 3; ++c) {
                        const int q = (int)std::lrint(std::fmin(std::fmax(v[c], 0.f), 65535.f));
                        row[(size_t)x * 6 + 2 * c] = (unsigned char)(q >> 8);
                        row[(size_t)x * 6 + 2 * c + 1] = (unsigned char)(q & 255);
                    }
                }
                o.write(reinterpret_cast<char *>(row.data()), (std::streamsize)row.size());
            }
        }
        std::printf("{\"width\": %d, \"height\": %d, \"out_width\": %d, \"out_height\": %d, \"total_ms_incl_io\": %.3f, \"mean\": %.4f}\n",
                    W, H, fw, fh, ms, sum / (3.0 * r.size()));
    } catch (const std::exception &e) {
        std::fprintf(stderr, "artgpu-cli: %s\n", e.what());
        return 1;
    }
    return 0;
}
