/*
 * include/artgpu.h -- C ABI of libartgpu.so: the MI355X (gfx950) implementation of ART's
 * raw-development hot path.  Plain pointers and sizes only; no C++ or torch types.
 *
 * The reference (artpixls/ART) has no FFI or plugin seam: the boundary this ABI replaces is
 * the C++ member/free-function layer of rtengine.  Each entry point names the reference
 * function whose body it stands in for; INTEGRATION.md shows the adapter a maintainer adds.
 *
 * Conventions
 *   - every function returns 0 on success, a negative ARTGPU_E* code otherwise;
 *     artgpu_last_error(ctx) gives the message.  The reference's functions return void and
 *     cannot fail (allocation failures are compiled out, rtengine/FTblockDN.cc:859); an
 *     adapter falls through to the CPU code on a non-zero return.
 *   - images are caller-owned.  artgpu_plane.on_device != 0 means `p` is a device (HBM)
 *     pointer usable on ctx's device; otherwise it is host memory and the call stages it
 *     through the context's device buffers (H2D before, D2H after).
 *   - a context is bound to one HIP device and one stream; calls on one context are
 *     serialised by the caller (the reference runs one pipeline thread per ImageProcessor,
 *     rtengine/improccoordinator.cc:192).  Device-pointer calls are asynchronous on the
 *     context's stream; host-pointer calls return after the D2H copy has completed.
 *   - there is no CPU fallback inside the library.
 */
#ifndef ARTGPU_H
#define ARTGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARTGPU_OK            0
#define ARTGPU_EINVAL       -1  /* bad argument */
#define ARTGPU_EHIP         -2  /* HIP runtime error (message has the hipError string) */
#define ARTGPU_ENOMEM       -3  /* device allocation failed */
#define ARTGPU_EUNSUPPORTED -4  /* valid in the reference, not implemented on the device path */

/* RAWParams::BayerSensor::Method values handled here (rtengine/procparams.h; dispatch at
 * rtengine/rawimagesource.cc:1862-1912). */
#define ARTGPU_BAYER_AMAZE 0
#define ARTGPU_BAYER_RCD   1

typedef struct artgpu_ctx artgpu_ctx;

/* One fp32 plane.  Mirrors array2D<float> (rtengine/array2D.h:74-296: row pointers into one
 * block, row stride W*4 bytes) and one channel of PlanarRGBData<float>
 * (rtengine/iimage.h:653-720: row stride = ceil16(W*4) bytes). */
typedef struct {
    float   *p;
    int32_t  w, h;
    int64_t  row_stride_bytes;
    int32_t  on_device;
} artgpu_plane;

typedef struct { artgpu_plane r, g, b; } artgpu_rgb;

/* Per-call device timings in milliseconds (hipEvents on the context's stream); filled only
 * when timing is enabled with artgpu_enable_timing(). */
typedef struct {
    float demosaic_ms;
    float border_ms;
    float total_ms;
} artgpu_timings;

int artgpu_create(int hip_device, artgpu_ctx **out);
int artgpu_destroy(artgpu_ctx *ctx);
const char *artgpu_last_error(const artgpu_ctx *ctx);
const char *artgpu_version(void);

/* Launch all work of this context on `hip_stream` (a hipStream_t; NULL = default stream). */
int artgpu_set_stream(artgpu_ctx *ctx, void *hip_stream);
int artgpu_synchronize(artgpu_ctx *ctx);
int artgpu_enable_timing(artgpu_ctx *ctx, int enable);
int artgpu_get_timings(const artgpu_ctx *ctx, artgpu_timings *out);

/* Replaces the Bayer branch of RawImageSource::demosaic (rtengine/rawimagesource.cc:1854-1912):
 *   ARTGPU_BAYER_AMAZE -> amaze_demosaic_RT(0,0,W,H,rawData,red,green,blue)
 *                         (rtengine/amaze_demosaic_RT.cc:41-1595) including its
 *                         border_interpolate2(W,H,3,...) when border < 4 (L1587-1589);
 *   ARTGPU_BAYER_RCD   -> rcd_demosaic() (rtengine/rcd_demosaic.cc:51-347) including its
 *                         border_interpolate2(W,H,9,...) (L342).
 * raw      : the CFA plane (RawImageSource::rawData), values 0..65535
 * filters  : RawImage::filters bit pattern (rtengine/rawimage.h:186-189), RGB Bayer only
 * initial_gain : RawImageSource::initialGain (clip_pt = 1/initialGain, amaze L53-54)
 * border   : RawImageSource::border (simpleprocess.cc:138-146)
 * out      : red, green, blue (same w,h as raw), fully overwritten. */
int artgpu_demosaic_bayer(artgpu_ctx *ctx, int method, const artgpu_plane *raw, uint32_t filters,
                          double initial_gain, int border, artgpu_rgb *out);

/* border_interpolate2 alone (rtengine/demosaic_algos.cc:200-353). */
int artgpu_border_interpolate2(artgpu_ctx *ctx, const artgpu_plane *raw, uint32_t filters,
                               int lborders, artgpu_rgb *out);

/* Bytes of device scratch the context currently holds (arena + staging). */
size_t artgpu_scratch_bytes(const artgpu_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* ARTGPU_H */
