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
void oracle_convert_color_space(float *const img[3], size_t s, int w, int h, const double mat[9]);
void oracle_exposure(float *const img[3], size_t s, int w, int h, float exp_scale, float black);
void oracle_filmlike_clip(float *const img[3], size_t s, int w, int h, float whitept);
float oracle_lutf(const float *data, int size, float index);
void oracle_tone_curve_std(float *const img[3], size_t s, int w, int h, const float *lut65536);

#ifdef __cplusplus
}
#endif
#endif
