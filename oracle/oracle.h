/*
 * oracle/oracle.h -- entry points of the CPU oracle (liboracle.so).
 * TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg are the only callers.  The product (libartgpu.so) never links or loads this.
 */
#ifndef ART_ORACLE_H
#define ART_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* strides are in floats */
int oracle_rcd_demosaic(const float *raw, size_t raw_stride, int W, int H, unsigned filters,
                        float *red, float *green, float *blue, size_t out_stride);
void oracle_rcd_tile(const float *raw, size_t rs, int W, int H, unsigned filters,
                     int tr, int tc, int numTh, int numTw,
                     float *red, float *green, float *blue, size_t os, float *work);
void oracle_border_interpolate2(int W, int H, int bord, const float *raw, size_t rs, unsigned filters,
                                float *red, float *green, float *blue, size_t os);

size_t oracle_amaze_arena_floats(void);
void oracle_amaze_tile(const float *raw, size_t rs, int width, int height, unsigned filters,
                       float clip_pt, float clip_pt8, int top, int left,
                       float *red, float *green, float *blue, size_t os,
                       float *arena, int poison);
int oracle_amaze_demosaic(const float *raw, size_t raw_stride, int W, int H, unsigned filters,
                          double initialGain, int border,
                          float *red, float *green, float *blue, size_t out_stride);

/* per-pixel stages (oracle/pixelops.c); strides in floats */
void oracle_get_image(const float *const src[3], size_t ss, int sx1, int sy1,
                      float *const dst[3], size_t ds, int w, int h, const float mul[3], int do_clip);
void oracle_get_image_skip(const float *const src[3], size_t ss, int W, int H, int sx1, int sy1, int skip,
                           float *const dst[3], size_t ds, int w, int h, const float mul[3], int do_clip);
void oracle_convert_color_space(float *const img[3], size_t s, int w, int h, const double mat[9]);
void oracle_exposure(float *const img[3], size_t s, int w, int h, float exp_scale, float black);
void oracle_filmlike_clip(float *const img[3], size_t s, int w, int h, float whitept);
float oracle_lutf(const float *data, int size, float index);
float oracle_xcbrtf(float d);
float oracle_xatan2f(float y, float x);
void oracle_xsincosf(float d, float *sn, float *cs);
void oracle_t_xatan2f(const float *y, const float *x, float *r, size_t n);
void oracle_t_xsincosf(const float *d, float *sn, float *cs, size_t n);
/* X-Trans (xtrans.c) */
void oracle_xtrans_demosaic(const float *raw, int width, int height, const int xtrans[36], const float rgb_cam[12], int passes, int use_cielab,
                            float *red, float *green, float *blue);
void oracle_xtrans_border(const float *raw, int width, int height, const int xtrans[36], int border, float *red, float *green, float *blue);
/* NEUTRAL tone curve (tonecurve.c) */
typedef struct {
    float ws[9], iws[9];            /* working space <-> XYZ, float casts (curves.cc:861-868) */
    float to_out[9], to_work[9];    /* output-profile gamut matrices (curves.cc:870-878) */
    float rhue, bhue, yhue, rrange, brange, yrange;   /* curves.cc:880-890 */
} oracle_neutral_state;
void oracle_pq_luts(float *pq65536, float *pq_inv65536);
void oracle_neutral_state_init(oracle_neutral_state *st, const double ws[9], const double iws[9], const float *to_out, const float *to_work);
void oracle_tone_curve_neutral(float *const img[3], size_t s, int w, int h, const float *lut65536, float whitecoeff,
                               const oracle_neutral_state *st, unsigned char *out_of_lut_range);
void oracle_t_xcbrtf(const float *x, float *y, size_t n);
int oracle_flat_curve_sample(const double *pts, int npts, int periodic, int ppn, double identity, int nout, double *out);
float oracle_noise_curve(const double *pts, int npts, float lut[501]);
void oracle_cachef(float lut[65536]);
void oracle_cachefy(float lut[65536]);
void oracle_denoise_gamma_tabs(float *gtab, float *igtab);
float oracle_lutf_noclip(const float *data, int size, float index);
void oracle_rgb2lab(float R, float G, float B, float *l, float *a, float *b, const float ws[9]);
void oracle_lab2rgb(float l, float a, float b, float *R, float *G, float *B, const float iws[9]);
void oracle_chroma_noise_map(const float *const img[3], size_t s, int w, int h, const double *mat, const float wpi[9],
                             const float curve[501], float *out);
void oracle_tone_curve_std(float *const img[3], size_t s, int w, int h, const float *lut65536);
extern int oracle_curve_tail_kind;      /* test hook, see pixelops.c */
extern double oracle_curve_tail_y;
float oracle_set_lut_val(const float *lut65536, float val);
void oracle_scale_colors(const void *src, int src_u16, int w, int h, const int cfa36[36], int bayer, const float cblacksom[4],
                         const float scale_mul[4], float *dst, float chmax[4]);
void oracle_channel_mixer(float *const img[3], size_t s, int w, int h, const float m[9]);
int oracle_rgb2out_matrix(const float *const src[3], float *const dst[3], size_t s, int w, int h, const float m[9], int linear, const float *lut, int lutsz);
void oracle_float_to_half(const float *x, unsigned short *y, size_t n);   /* DNG_FloatToHalf, halffloat.h:9-46 */
void oracle_get_scanlines(const float *const img[3], size_t s, int w, int h, int bps, int is_float, void *out);
void oracle_hsl_equalizer(float *const img[3], int W, int H, const double *hcurve, int nh, const double *scurve, int ns, const double *lcurve, int nl,
                          int smoothing, const double ws[9], double scale, int to_rgb);
void oracle_image_rgb_to_lab(float *const img[3], int W, int H, const double ws[9]);
void oracle_image_lab_to_rgb(float *const img[3], int W, int H, const double iws[9]);
void oracle_lab_histogram(const float *L, int W, int H, unsigned hist[65536]);
void oracle_lab_adjustments(float *const img[3], int W, int H, const float *lcurve, const float *acurve, const float *bcurve, float chroma);
void oracle_rgb2l(const float *R, const float *G, const float *B, float *L, int W, int H);
float oracle_calc_contrast_threshold(const float *L, int W, int tileY, int tileX, int ts, float factor);
float oracle_build_blend_mask(const float *L, float *blend, int W, int H, float contrastThreshold, int autoContrast);
void oracle_bayer_bilinear_blend(const float *blend, const float *raw, float *red, float *green, float *blue, int W, int H, unsigned filters);
void oracle_dual_demosaic_blend2(const float *raw, float *red, float *green, float *blue, int W, int H, unsigned filters, double *contrast, int autoContrast, int vng4);
void oracle_dual_demosaic_blend(const float *raw, float *red, float *green, float *blue, int W, int H, unsigned filters, double *contrast, int autoContrast);
unsigned oracle_prefilters(unsigned filters);
int oracle_vng4_code(unsigned pf, int width, int row, int col, int32_t *ip0);
void oracle_vng4_demosaic(const float *raw, int W, int H, unsigned filters, unsigned prefilters, float *red, float *green, float *blue);
float oracle_logenc_find_gray(float source_gray, float target_gray);
void oracle_log_encoding(float *const img[3], int W, int H, const double ws[9], double gain, double targetGray, double blackEv, double whiteEv,
                         int regularization, int satcontrol, int highlightCompression, int full_width, int full_height);
void oracle_saturation_vibrance(float *const img[3], size_t s, int w, int h, int saturation_p, int vibrance_p, const double ws[9]);
void oracle_rgb_curves(float *const img[3], size_t s, int w, int h, const float *const luts[3]);
/* AUTOMATIC chrominance estimation (oracle/dninfo.c) */
float oracle_autodn_adjust(float *chaut_io, int Nb, float maxmax, float lumema, float chromina, float redyel, float skinc, float nsknc, int aggressive);
void oracle_denoise_info_crop(const float *const crop[3], int crW, int crH, const double mat[9], const float wp[9], double gamma, int aggressive, float *info);
int oracle_denoise_compute_params(const float *const planes[3], size_t ss, int W, int H, int border, const float mul[3], int do_clip,
                                  const double mat[9], const float wp[9], double gamma, int aggressive, float *store_out, float *info_out);
void oracle_rgb_to_yuv(float *const img[3], size_t s, int w, int h, const float ws[9]);
void oracle_yuv_to_rgb(float *const img[3], size_t s, int w, int h, const float ws[9]);

/* wavelet_decomposition, subsampling == 1 (oracle/wavelet.c) */
typedef struct {
    int w, h, w2, h2, nlevels;
    float *band[10][4]; /* band[l][1..3]: detail subbands, w2*h2 floats each */
    float *coeff0;      /* final low-pass, w2*h2 */
} oracle_wavelet;
int oracle_wavelet_skip(int level);
oracle_wavelet *oracle_wavelet_decompose(const float *src, int w, int h, int maxlvl);
void oracle_wavelet_reconstruct(oracle_wavelet *d, float *dst, float blend);
void oracle_wavelet_free(oracle_wavelet *d);

/* wavelet part of RGB_denoise (oracle/denoise.c) */
typedef struct {
    double luminance, luminanceDetail, chrominance, chrominanceRedGreen, chrominanceBlueYellow, gamma;
    double expcomp;   /* RGB_denoise's expcomp argument (0 when called from ImProcFunctions::denoise) */
    double scale;     /* ImProcData::scale (1 for full-size export) */
    int autoch;       /* chrominanceMethod == AUTOMATIC */
    int aggressive;   /* DenoiseParams::aggressive -> QUALITY_HIGH */
    int detail_thresh; /* DenoiseParams::luminanceDetailThreshold */
    int lab_mode;      /* DenoiseParams::colorSpace == LAB */
    float iws[9];      /* working-space inverse matrix (LAB mode only) */
} oracle_denoise_params;
float oracle_madrgb(const float *data, int datalen);
void oracle_boxblur_flat(const float *src, float *dst, float *temp, int radx, int rady, int W, int H);
void oracle_shrink_all_L(oracle_wavelet *L, int level, int dir, const float *noisevarlum, const float *madL3, double scale);
void oracle_shrink_all_AB(const oracle_wavelet *L, oracle_wavelet *ab, int level, int dir, const float *noisevarchrom,
                          float noisevar_ab, int useNoiseCCurve, int autoch, const float *madL3, double scale);
void oracle_bishrink_AB(const oracle_wavelet *L, oracle_wavelet *ab, const float *noisevarchrom, float noisevar_ab, int useNoiseCCurve,
                        int autoch, float madL[8][3], double scale);
void oracle_gamma_lut(float *lut, float gamma, float start, float slope, float divisor, float factor);
int oracle_rgb_denoise(float *const img[3], size_t stride, int w, int h, const oracle_denoise_params *p,
                       const float wpi[9], const float *noisevarchrom_in, float *Lin_out, float *Lden_out, int detail_recovery);

/* DCT detail recovery (oracle/detail.c) */
void oracle_detail_tilemasks(float *tilemask_in, float *tilemask_out);
float oracle_detail_factor(float d);
void oracle_detail_recovery(int width, int height, float *L, const float *Lin, float params_Ldetail, double scale);
void oracle_detail_recovery_ex(int width, int height, float *L, const float *Lin, float params_Ldetail, double scale, int detail_thresh);

/* guided chroma smoothing (oracle/guided.c) */
void oracle_boxblur_ring(float *img, int radius, int W, int H);
float oracle_bilinear(const float *src, int W, int H, float x, float y);
void oracle_rescale_bilinear(const float *src, int Ws, int Hs, float *dst, int Wd, int Hd);
void oracle_guided_filter(const float *guide, const float *src, float *dst, int W, int H, int r, float epsilon, int subsampling);
void oracle_guided_filter_log(const float *guide, float base, float *chan, int W, int H, int r, float eps, int subsampling);
void oracle_denoise_guided_smoothing(float *const img[3], int W, int H, const double ws[9], int guidedChromaRadius, double scale);

/* NL-means stage and helpers (oracle/nlmeans.c) */
void oracle_yvv_factors(double sigma, double *b1, double *b2, double *b3, double *B, double M[9]);
void oracle_gaussian_blur(float *img, int W, int H, double sigma);
void oracle_detail_mask(const float *src, float *mask, int W, int H, float scaling, float threshold, float ceiling, float factor, float blur);
float oracle_lutf_vec(const float *data, int size, float index);
void oracle_nlmeans(float *img, int W, int H, float normcoeff, int strength, int detail_thresh, float scale);

/* sleef-derived math (oracle/sleef.c); _s = scalar form, _v = per-lane SSE form */
float oracle_xexpf_s(float d);
float oracle_xexpf_v(float d);
float oracle_xexpf_v_nocheck(float d);
float oracle_xlogf_s(float d);
float oracle_xlogf_v(float d);
float oracle_xlogf_v_nocheck(float d);
float oracle_pow_F(float a, float b);
float oracle_xlin2log(float x, float base);
float oracle_xlog2lin(float x, float base);

#ifdef __cplusplus
}
#endif
#endif
