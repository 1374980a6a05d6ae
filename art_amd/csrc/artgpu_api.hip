// art_amd/csrc/artgpu_api.hip -- the C ABI of libartgpu.so (include/artgpu.h): context,
// device buffers, staging for host-pointer calls, kernel launches, error reporting.
// There is no CPU fallback here: every entry point either runs the HIP kernels or fails.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>

#include "../../include/artgpu.h"
#include "kernels.h"

using namespace artgpu;

struct artgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t aux = nullptr;      // second stream (serial statistics of the AUTOMATIC chroma estimation run beside the decompositions)
    hipEvent_t aux_ev[2] = {nullptr, nullptr};
    // RGB_denoise: the DCT detail recovery of L runs on a side stream beside the chroma blurs / reconstructions
    hipStream_t dn_stream[1] = {nullptr};
    hipEvent_t dn_ev[2] = {nullptr, nullptr};
    const float *gam_tab = nullptr; float gam_key[6] = {};   // RGB_denoise's gamma / inverse-gamma tables in pool[P_GAM]: what they were built from
    int opt_lut_lds = 1;           // 0: never the LUT-in-LDS shapes of the pixel passes (tests compare the two)
    int opt_dn_streams = 0;        // 1: the DCT detail recovery of L on a side stream beside the reconstructions of a and b.  The default until round 5, when the
                                   // stage waited on LDS round trips and left the chip half idle; since detail_blocks_kernel lost a fifth of its time (detail.hip) the
                                   // two chains only get in each other's way: one kernel after the other is 0.1 - 0.15 ms per 45 MP frame faster (scripts/r5_ab9.sh)
    int ccalc_nonneg = 0;          // set by artgpu_improc_denoise around RGB_denoise: the chroma noise map is the one chroma_map_kernel has just written (squares: no negative value)
    int opt_dn_fused = 1;          // ShrinkAllL / ShrinkAllAB -- 0: three kernels per channel (factors, row sums, column sums + update); 2: one kernel per
                                   // channel; 1: one kernel, and one launch for all three channels where nothing has to happen between them
    std::string err;
    int *fs_diag = nullptr;        // pinned host words the fused shrink pass writes before it traps (which strip waited for which): see fail()
    int opt_dn_debug_stall = -1;   // test hook: band << 16 | strip of the fused shrink pass that never publishes its progress (-1: none)
    long opt_dn_wait_ms = 0;       // how long a strip of the fused shrink pass waits for the strip above before it gives up (0: five seconds)
    long long batch_px = 0;        // artgpu_batch_run: pixels of the largest frame this context's scratch was grown for since the last trim
    int dn_form = -1, dn_form_streak = 0;   // RGB_denoise: the shrink passes' form of the last call (1 fused / 0 three kernels) and how many calls in a row used it
    // per-workgroup work arenas (demosaic)
    float *arena = nullptr;
    size_t arena_bytes = 0;
    // staging for host-pointer calls: one CFA plane + three output planes
    static constexpr int NSTAGE = 8;
    float *stage[NSTAGE] = {};
    size_t stage_bytes[NSTAGE] = {};
    // grow-only scratch pool for the denoise path (planes, decompositions, shrink buffers)
    static constexpr int NPOOL = 64;
    float *pool[NPOOL] = {};
    size_t pool_bytes[NPOOL] = {};
    // artgpu_batch_run lanes: sibling contexts (own stream, arena, pools) that take every lanes-th frame on their own host thread
    std::vector<artgpu_ctx *> lanes;
    int batch_lanes = 1;
    int frames_in_flight = 1;      // set by artgpu_batch_run on itself and its lanes while a batch with L > 1 lanes runs: the demosaic then takes 5/8 of the CUs (option amaze_grid)
    bool owns_stream = false;
    // artgpu_batch_run_io: a frame's upload and download run on streams of their own beside the kernels on `stream`.  Events, by staging slot
    // (the parity of the frame's turn on this lane): [0,1] uploaded, [2,3] the upload slot has been read, [4,5] scanlines written, [6,7] downloaded
    hipStream_t io_up = nullptr, io_down = nullptr;
    hipEvent_t io_ev[8] = {};
    int *io_host = nullptr;        // pinned, 8 words per frame of the lane: channel maxima (bit patterns) x 3, -, rgb2out's count of unsupported values
    int io_host_cap = 0;
    int cu_reserve = 0;            // set around a batch whose downloads run as a kernel of a few workgroups: the persistent one-workgroup-per-CU pixel passes leave those CUs alone
    int opt_io_direct = -1;        // artgpu_batch_run_io, scanlines into pinned host memory: n > 0: written there by n persistent workgroups (no staging, no copy); 0: staged +
                                   // hipMemcpy; -1 (default): 8 with two lanes, 0 otherwise (io_frame has the measurements)
    float fuse_pre = 0.f, fuse_post = 0.f;   // improc_denoise -> rgb_denoise: exposure compensation fused into rgb2yuv / yuv2rgb
    GetImageFuse fuse_gi = {};               // improc_denoise_fused -> chroma map, rgb2yuv: getImage + matrix read from the demosaiced planes
    float fuse_exp_scale = 0.f, fuse_exp_black = 0.f;   // improc_denoise_fused -> yuv2rgb: ImProcFunctions::exposure behind the last pass
    int fuse_exp_on = 0;
    int tail_exp_on = 0;                      // improc_denoise_fused: the exposure rides on the tool's LAST pixel pass when that is not yuv2rgb
    float tail_exp_scale = 0.f, tail_exp_black = 0.f;
    float *bbox = nullptr; // AMaZE: per-tile nyquist bounding boxes
    size_t bbox_bytes = 0;
    // AMaZE v2: tile lists on the device (ints).  [stream tiles | arena-tile template: count, tiles | working copy: count, tiles + room
    // for every streamed tile the stream kernel hands back]
    float *amz_lists = nullptr;
    size_t amz_lists_bytes = 0;
    int amz_w = 0, amz_h = 0, amz_nstream = 0, amz_narena = 0, amz_mode = -1;
    int num_cus = 0;
    // options (artgpu_set_option): test / profiling switches that used to be environment variables
    int opt_amaze_path = 0;        // 0: LDS streaming kernel for full tiles + arena kernel for the rest; 1: arena kernel for every tile
    int opt_amaze_split = 0;       // 1 (with path 1): one launch per phase
    long opt_amaze_zero_mask = 0x81f0;
    int opt_amaze_zero_frame = 16;
    int opt_amaze_poison = -1;     // >= 0: byte pattern the arenas are filled with before the launch
    int opt_roctx = 0;             // 1: roctx ranges named after the reference functions around the entry points (rocprofv3 --marker-trace)
    artgpu_progress_fn progress_fn = nullptr;   // artgpu_set_progress_callback
    void *progress_user = nullptr;
    int opt_amaze_overlap = 1;     // 0: the arena kernel's static tiles behind the stream kernel instead of beside it
    int opt_amaze_grid = 0;        // > 0: at most this many stream workgroups (each owns a CU): with several frames in flight the CUs left over
                                   // take the other frames' bandwidth-bound passes while this frame's issue-bound demosaic runs
    hipStream_t amz_side = nullptr;
    hipEvent_t amz_ev[2] = {nullptr, nullptr};
    int opt_rcd_rows = 8;          // rows per iteration of the streaming kernel (4 or 8)
    int *rcd_counter = nullptr;    // RCD streaming kernel: tile counter
    int curve_tail_kind = ARTGPU_CURVE_TAIL_HOST;   // artgpu_set_curve_tail
    double curve_tail_y = 1.0;
    ParamCurve curve_tail_pc = {};
    float *lut = nullptr; // 65536-entry tone LUT on the device
    size_t lut_bytes = 0;
    std::vector<float> lut_host;           // what ctx->lut holds (a curve that comes back unchanged is not uploaded again)
    std::vector<float> ncurve_host;        // likewise the 501-entry chroma noise curve behind the cachef table
    char *tab_ring = nullptr;              // pinned staging ring of h2d_table (caller look-up tables)
    size_t tab_ring_bytes = 0, tab_ring_off = 0;
    bool tab_ring_busy = false;
    hipEvent_t tab_ev = nullptr;
    std::vector<float> rgbcurve_host[3];   // likewise the three rgbCurves tables (P_RGBCURVES: a slot nothing else writes)
    // timing
    bool timing = false;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    artgpu_timings last = {0.f, 0.f, 0.f};
};

namespace {

int fail(artgpu_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) {
        ctx->err = buf;
        // a fault raised by the fused shrink pass's bounded wait (shrinkblur.hip) is attributed: the kernel left these words in pinned host
        // memory when it gave up (check_async_faults finds them at the next synchronisation point)
        if (code == ARTGPU_EHIP && ctx->fs_diag && (unsigned)ctx->fs_diag[0] == 0xF5D1A600u) {
            char more[192];
            snprintf(more, sizeof more, " [shrink_blur_kernel: band %d strip %d gave up waiting for the strip above to hand down block %d (its counter: %d)]",
                     ctx->fs_diag[1], ctx->fs_diag[2], ctx->fs_diag[3], ctx->fs_diag[4]);
            ctx->err += more;
            ctx->fs_diag[0] = 0;      // reported once: a later, unrelated failure is not attributed to it
        }
    }
    return code;
}

#define HIPCHK(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(ctx, ARTGPU_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// A call that puts work on a side stream has to leave with the context's stream behind that work on EVERY path: an error return in between
// would otherwise leave the side stream running on pool buffers the next call hands out again.  Armed when the fork happens, disarmed by the
// regular join (an event wait, no host blocking); on any other exit the destructor drains the side stream.
struct SideStreamJoin {
    hipStream_t side = nullptr;
    void arm(hipStream_t s) { side = s; }
    void disarm() { side = nullptr; }
    ~SideStreamJoin() { if (side) (void)hipStreamSynchronize(side); }
};

int ensure(artgpu_ctx *ctx, float **buf, size_t *cur, size_t need)
{
    if (*cur >= need) return ARTGPU_OK;
    if (*buf) { HIPCHK(ctx, hipFree(*buf)); *buf = nullptr; *cur = 0; }
    hipError_t e = hipMalloc(reinterpret_cast<void **>(buf), need);
    if (e != hipSuccess) {
        *buf = nullptr;
        return fail(ctx, ARTGPU_ENOMEM, "hipMalloc(%zu bytes) failed: %s", need, hipGetErrorString(e));
    }
    *cur = need;
    return ARTGPU_OK;
}

// A caller's look-up table -> device memory, one lifetime rule for all of them (artgpu.h "Host look-up tables"): the array is free when
// the entry point returns, whatever kind of memory it is (pageable, pinned, registered, managed).  The table is copied into a pinned ring the
// context owns and travels from there on the context's stream, so the call neither depends on how HIP treats the caller's memory type nor
// drains the stream; the host only waits (for the ring's last copy) when the ring wraps, every dozen calls.
int h2d_table(artgpu_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (need > ctx->tab_ring_bytes) {
        if (ctx->tab_ring) {
            if (ctx->tab_ring_busy) HIPCHK(ctx, hipEventSynchronize(ctx->tab_ev));
            HIPCHK(ctx, hipHostFree(ctx->tab_ring));
            ctx->tab_ring = nullptr; ctx->tab_ring_bytes = 0; ctx->tab_ring_busy = false;
        }
        const size_t cap = need * 2 > ((size_t)8 << 20) ? need * 2 : ((size_t)8 << 20);
        if (hipHostMalloc(reinterpret_cast<void **>(&ctx->tab_ring), cap, hipHostMallocDefault) != hipSuccess) {
            ctx->tab_ring = nullptr;
            return fail(ctx, ARTGPU_ENOMEM, "pinned staging ring for look-up tables (%zu bytes)", cap);
        }
        ctx->tab_ring_bytes = cap; ctx->tab_ring_off = 0;
        if (!ctx->tab_ev) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->tab_ev, hipEventDisableTiming));
    }
    if (ctx->tab_ring_off + need > ctx->tab_ring_bytes) {        // wrap: what was staged before has to have left the ring
        if (ctx->tab_ring_busy) HIPCHK(ctx, hipEventSynchronize(ctx->tab_ev));
        ctx->tab_ring_busy = false;
        ctx->tab_ring_off = 0;
    }
    char *slot = ctx->tab_ring + ctx->tab_ring_off;
    std::memcpy(slot, src, bytes);
    HIPCHK(ctx, hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->tab_ev, ctx->stream));
    ctx->tab_ring_busy = true;
    ctx->tab_ring_off += need;
    return ARTGPU_OK;
}

bool plane_ok(const artgpu_plane *p)
{
    return p && p->p && p->w > 0 && p->h > 0 && p->row_stride_bytes >= (int64_t)p->w * 4 && (p->row_stride_bytes % 4) == 0;
}

struct DevImage {
    const float *raw; size_t raw_stride;
    float *r, *g, *b; size_t out_stride;
    bool staged;
};

// Resolve device pointers for (raw, out); host planes are staged through ctx buffers.
int bind_images(artgpu_ctx *ctx, const artgpu_plane *raw, artgpu_rgb *out, DevImage *d)
{
    const int W = raw->w, H = raw->h;
    const artgpu_plane *op[3] = {&out->r, &out->g, &out->b};
    for (int k = 0; k < 3; ++k) {
        if (!plane_ok(op[k]) || op[k]->w != W || op[k]->h != H) return fail(ctx, ARTGPU_EINVAL, "output plane %d: bad pointer/size/stride", k);
        if ((op[k]->on_device != 0) != (out->r.on_device != 0) || op[k]->row_stride_bytes != out->r.row_stride_bytes)
            return fail(ctx, ARTGPU_EINVAL, "output planes must share residency and row stride");
    }
    d->staged = false;
    if (raw->on_device) {
        d->raw = raw->p;
        d->raw_stride = (size_t)(raw->row_stride_bytes / 4);
    } else {
        int rc = ensure(ctx, &ctx->stage[0], &ctx->stage_bytes[0], (size_t)W * H * 4);
        if (rc) return rc;
        HIPCHK(ctx, hipMemcpy2DAsync(ctx->stage[0], (size_t)W * 4, raw->p, (size_t)raw->row_stride_bytes, (size_t)W * 4, H, hipMemcpyHostToDevice, ctx->stream));
        d->raw = ctx->stage[0];
        d->raw_stride = W;
    }
    if (out->r.on_device) {
        d->r = out->r.p; d->g = out->g.p; d->b = out->b.p;
        d->out_stride = (size_t)(out->r.row_stride_bytes / 4);
    } else {
        for (int k = 0; k < 3; ++k) {
            int rc = ensure(ctx, &ctx->stage[1 + k], &ctx->stage_bytes[1 + k], (size_t)W * H * 4);
            if (rc) return rc;
        }
        d->r = ctx->stage[1]; d->g = ctx->stage[2]; d->b = ctx->stage[3];
        d->out_stride = W;
        d->staged = true;
    }
    return ARTGPU_OK;
}

int unbind_images(artgpu_ctx *ctx, artgpu_rgb *out, const DevImage *d)
{
    if (!d->staged) return ARTGPU_OK;
    const int W = out->r.w, H = out->r.h;
    artgpu_plane *op[3] = {&out->r, &out->g, &out->b};
    const float *src[3] = {d->r, d->g, d->b};
    for (int k = 0; k < 3; ++k)
        HIPCHK(ctx, hipMemcpy2DAsync(op[k]->p, (size_t)op[k]->row_stride_bytes, src[k], (size_t)W * 4, (size_t)W * 4, H, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ARTGPU_OK;
}

int launch_border(artgpu_ctx *ctx, const DevImage &d, int W, int H, unsigned filters, int bord)
{
    if (bord <= 0) return ARTGPU_OK;
    if (W <= 2 * bord || H <= 2 * bord) return fail(ctx, ARTGPU_EUNSUPPORTED, "border_interpolate2: image %dx%d too small for border %d", W, H, bord);
    BorderArgs b;
    b.raw = d.raw; b.raw_stride = d.raw_stride;
    b.red = d.r; b.green = d.g; b.blue = d.b; b.out_stride = d.out_stride;
    b.W = W; b.H = H; b.bord = bord; b.filters = filters;
    const long long total = 2LL * bord * H + 2LL * bord * (W - 2 * bord);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    HIPCHK(ctx, launch_border_interpolate2(b, grid, ctx->stream));
    return ARTGPU_OK;
}


// Generic RGB image binding: device planes are used in place, host planes are staged through
// ctx->stage[slot..slot+2] (copy_in: H2D now; the D2H happens in unbind_rgb).
struct DevRGB {
    float *p[3];
    size_t stride; // floats
    int w, h;
    bool staged;
};

int bind_rgb(artgpu_ctx *ctx, const artgpu_rgb *img, int slot, bool copy_in, DevRGB *d, const char *what)
{
    const artgpu_plane *pl[3] = {&img->r, &img->g, &img->b};
    for (int k = 0; k < 3; ++k) {
        if (!plane_ok(pl[k]) || pl[k]->w != img->r.w || pl[k]->h != img->r.h)
            return fail(ctx, ARTGPU_EINVAL, "%s: plane %d has a bad pointer/size/stride", what, k);
        if ((pl[k]->on_device != 0) != (img->r.on_device != 0) || pl[k]->row_stride_bytes != img->r.row_stride_bytes)
            return fail(ctx, ARTGPU_EINVAL, "%s: planes must share residency and row stride", what);
    }
    d->w = img->r.w; d->h = img->r.h;
    if (img->r.on_device) {
        for (int k = 0; k < 3; ++k) d->p[k] = pl[k]->p;
        d->stride = (size_t)(img->r.row_stride_bytes / 4);
        d->staged = false;
        return ARTGPU_OK;
    }
    const size_t rowb = (size_t)d->w * 4;
    for (int k = 0; k < 3; ++k) {
        int rc = ensure(ctx, &ctx->stage[slot + k], &ctx->stage_bytes[slot + k], rowb * d->h);
        if (rc) return rc;
        d->p[k] = ctx->stage[slot + k];
        if (copy_in)
            HIPCHK(ctx, hipMemcpy2DAsync(d->p[k], rowb, pl[k]->p, (size_t)pl[k]->row_stride_bytes, rowb, d->h, hipMemcpyHostToDevice, ctx->stream));
    }
    d->stride = d->w;
    d->staged = true;
    return ARTGPU_OK;
}

int unbind_rgb(artgpu_ctx *ctx, artgpu_rgb *img, const DevRGB *d)
{
    if (!d->staged) return ARTGPU_OK;
    artgpu_plane *pl[3] = {&img->r, &img->g, &img->b};
    const size_t rowb = (size_t)d->w * 4;
    for (int k = 0; k < 3; ++k)
        HIPCHK(ctx, hipMemcpy2DAsync(pl[k]->p, (size_t)pl[k]->row_stride_bytes, d->p[k], rowb, rowb, d->h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ARTGPU_OK;
}

#ifndef ARTGPU_MAX_TILE_WG
#define ARTGPU_MAX_TILE_WG 8192
#endif
constexpr int MAX_TILE_WORKGROUPS = ARTGPU_MAX_TILE_WG;

} // namespace

extern "C" {

const char *artgpu_version(void) { return "artgpu 0.1 (gfx950)"; }

int artgpu_create(int hip_device, artgpu_ctx **out)
{
    if (!out) return ARTGPU_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || hip_device < 0 || hip_device >= n) return ARTGPU_EHIP;
    artgpu_ctx *ctx = new (std::nothrow) artgpu_ctx;
    if (!ctx) return ARTGPU_ENOMEM;
    ctx->device = hip_device;
    if (hipSetDevice(hip_device) != hipSuccess) { delete ctx; return ARTGPU_EHIP; }
    for (int k = 0; k < 3; ++k)
        if (hipEventCreate(&ctx->ev[k]) != hipSuccess) {
            for (int j = 0; j < k; ++j) (void)hipEventDestroy(ctx->ev[j]);     // the events created so far
            delete ctx;
            return ARTGPU_EHIP;
        }
    *out = ctx;
    return ARTGPU_OK;
}

int artgpu_destroy(artgpu_ctx *ctx)
{
    if (!ctx) return ARTGPU_EINVAL;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->arena) (void)hipFree(ctx->arena);
    for (int k = 0; k < artgpu_ctx::NSTAGE; ++k)
        if (ctx->stage[k]) (void)hipFree(ctx->stage[k]);
    if (ctx->lut) (void)hipFree(ctx->lut);
    if (ctx->tab_ring) (void)hipHostFree(ctx->tab_ring);
    if (ctx->fs_diag) (void)hipHostFree(ctx->fs_diag);
    if (ctx->tab_ev) (void)hipEventDestroy(ctx->tab_ev);
    if (ctx->bbox) (void)hipFree(ctx->bbox);
    if (ctx->amz_lists) (void)hipFree(ctx->amz_lists);
    if (ctx->rcd_counter) (void)hipFree(ctx->rcd_counter);
    for (int k = 0; k < artgpu_ctx::NPOOL; ++k)
        if (ctx->pool[k]) (void)hipFree(ctx->pool[k]);
    for (int k = 0; k < 3; ++k)
        if (ctx->ev[k]) (void)hipEventDestroy(ctx->ev[k]);
    for (artgpu_ctx *l : ctx->lanes) (void)artgpu_destroy(l);
    ctx->lanes.clear();
    if (ctx->io_up) { (void)hipStreamSynchronize(ctx->io_up); (void)hipStreamDestroy(ctx->io_up); }
    if (ctx->io_down) { (void)hipStreamSynchronize(ctx->io_down); (void)hipStreamDestroy(ctx->io_down); }
    for (int k = 0; k < 8; ++k)
        if (ctx->io_ev[k]) (void)hipEventDestroy(ctx->io_ev[k]);
    if (ctx->io_host) (void)hipHostFree(ctx->io_host);
    if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    for (int k = 0; k < 2; ++k)
        if (ctx->aux_ev[k]) (void)hipEventDestroy(ctx->aux_ev[k]);
    if (ctx->aux) { (void)hipStreamSynchronize(ctx->aux); (void)hipStreamDestroy(ctx->aux); }
    for (int k = 0; k < 2; ++k)
        if (ctx->dn_ev[k]) (void)hipEventDestroy(ctx->dn_ev[k]);
    if (ctx->dn_stream[0]) { (void)hipStreamSynchronize(ctx->dn_stream[0]); (void)hipStreamDestroy(ctx->dn_stream[0]); }
    for (int k = 0; k < 2; ++k)
        if (ctx->amz_ev[k]) (void)hipEventDestroy(ctx->amz_ev[k]);
    if (ctx->amz_side) { (void)hipStreamSynchronize(ctx->amz_side); (void)hipStreamDestroy(ctx->amz_side); }
    delete ctx;
    return ARTGPU_OK;
}

const char *artgpu_last_error(const artgpu_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int artgpu_set_stream(artgpu_ctx *ctx, void *hip_stream)
{
    if (!ctx) return ARTGPU_EINVAL;
    hipStream_t next = static_cast<hipStream_t>(hip_stream);
    if (next == ctx->stream) return ARTGPU_OK;
    // The context's scratch planes and the tables it keeps between calls (curves, gamma / Lab / DCT tables) were last written on the
    // stream it is leaving: the new stream's work is ordered behind that.  (A previous stream that no longer exists cannot be waited
    // for -- its work is done by definition -- so a failure here is not an error.)
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
        if (hipEventRecord(ev, ctx->stream) == hipSuccess) (void)hipStreamWaitEvent(next, ev, 0);
        (void)hipEventDestroy(ev);
    }
    (void)hipGetLastError();
    ctx->stream = next;
    return ARTGPU_OK;
}

// A kernel that gave up a bounded wait (the fused shrink pass: a strip whose predecessor did not hand down a block within five seconds) says so in
// pinned host words and runs to its end with wrong coefficients instead of trapping -- a trap aborts the host process inside the runtime.  The
// library reads the words wherever it has just waited for the stream and turns them into ARTGPU_EHIP with the band / strip / block in
// artgpu_last_error (fail() appends them); the frame in flight is invalid, the context stays usable.
static int check_async_faults(artgpu_ctx *ctx)
{
    if (ctx->fs_diag && (unsigned)ctx->fs_diag[0] == 0xF5D1A600u) return fail(ctx, ARTGPU_EHIP, "a kernel gave up a bounded wait: the results of the calls since the last synchronisation are invalid");
    return ARTGPU_OK;
}

int artgpu_synchronize(artgpu_ctx *ctx)
{
    if (!ctx) return ARTGPU_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return check_async_faults(ctx);
}

int artgpu_set_curve_tail(artgpu_ctx *ctx, int kind, double y_last)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (kind < ARTGPU_CURVE_TAIL_LUT || kind > ARTGPU_CURVE_TAIL_HOST) return fail(ctx, ARTGPU_EINVAL, "set_curve_tail: kind %d (a parametric curve: artgpu_set_curve_tail_parametric)", kind);
    ctx->curve_tail_kind = kind;
    ctx->curve_tail_y = y_last;
    return ARTGPU_OK;
}

int artgpu_set_curve_tail_parametric(artgpu_ctx *ctx, const double *p, int np)
{
    if (!ctx) return ARTGPU_EINVAL;
    // DiagonalCurve's constructor (diagonalcurves.cc:106-131): eight or nine parameters behind the kind; a curve whose four slider values
    // are all zero is the identity there (and `identity` curves never reach setLutVal with a Curve object: ARTGPU_CURVE_TAIL_LUT)
    if (!p || (np != 8 && np != 9)) return fail(ctx, ARTGPU_EINVAL, "set_curve_tail_parametric: 8 or 9 parameters (p[0] = DCT_Parametric)");
    if (p[4] == 0.0 && p[5] == 0.0 && p[6] == 0.0 && p[7] == 0.0) return fail(ctx, ARTGPU_EINVAL, "set_curve_tail_parametric: an identity curve has no Curve object (ARTGPU_CURVE_TAIL_LUT)");
    pc_init(ctx->curve_tail_pc, p, np);
    ctx->curve_tail_kind = ARTGPU_CURVE_TAIL_PARAMETRIC;
    return ARTGPU_OK;
}

int artgpu_get_option(artgpu_ctx *ctx, const char *name, long *value)
{
    if (!ctx || !name || !value) return ARTGPU_EINVAL;
    const std::string n(name);
    if (n.rfind("amaze_counter", 0) == 0 && n.size() == 14 && n[13] >= '0' && n[13] <= '7') {
        // counters of the last AMaZE call (0 entries pulled by stream workgroups, 1 pulls that found nothing, 2 entries published,
        // 3 tiles handed to the arena list, 4-6 what the arena kernel saw: listed tiles, queue taken, queue reserved)
        if (!ctx->amz_lists || ctx->amz_nstream <= 0) { *value = 0; return ARTGPU_OK; }
        HIPCHK(ctx, hipSetDevice(ctx->device));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        const size_t nfixed = (((size_t)ctx->amz_nstream + (1 + ctx->amz_narena) + (1 + ctx->amz_narena + ctx->amz_nstream) + 3) / 4) * 4;
        int v = 0;
        HIPCHK(ctx, hipMemcpy(&v, reinterpret_cast<int *>(ctx->amz_lists) + nfixed + 4 + 2 * (size_t)ctx->amz_nstream + (n[13] - '0'), sizeof(int), hipMemcpyDeviceToHost));
        *value = v;
        return ARTGPU_OK;
    }
    return fail(ctx, ARTGPU_EINVAL, "get_option: unknown option '%s'", name);
}

int artgpu_set_option(artgpu_ctx *ctx, const char *name, long value)
{
    if (!ctx || !name) return ARTGPU_EINVAL;
    const std::string n(name);
    if (n == "amaze_path") { if (value < 0 || value > 1) return fail(ctx, ARTGPU_EINVAL, "amaze_path: 0 or 1"); ctx->opt_amaze_path = (int)value; }
    else if (n == "amaze_split") ctx->opt_amaze_split = value != 0;
    else if (n == "amaze_overlap") ctx->opt_amaze_overlap = value != 0;
    else if (n == "amaze_grid") { if (value < 0) return fail(ctx, ARTGPU_EINVAL, "amaze_grid: >= 0"); ctx->opt_amaze_grid = (int)value; }
    else if (n == "amaze_zero_mask") ctx->opt_amaze_zero_mask = value;
    else if (n == "amaze_zero_frame") ctx->opt_amaze_zero_frame = (int)value;
    else if (n == "amaze_poison") ctx->opt_amaze_poison = (int)value;
    else if (n == "roctx") ctx->opt_roctx = value != 0;
    else if (n == "dn_streams") ctx->opt_dn_streams = value != 0;
    else if (n == "dn_fused") ctx->opt_dn_fused = (int)value;
    else if (n == "dn_debug_stall") ctx->opt_dn_debug_stall = (int)value;
    else if (n == "dn_wait_ms") ctx->opt_dn_wait_ms = value < 0 ? 0 : value;
    else if (n == "lut_lds") ctx->opt_lut_lds = value != 0;
    else if (n == "cu_reserve") { if (value < 0 || value > 4096) return fail(ctx, ARTGPU_EINVAL, "cu_reserve: 0 .. 4096"); ctx->cu_reserve = (int)value; }
    else if (n == "io_direct") { if (value < -1 || value > 4096) return fail(ctx, ARTGPU_EINVAL, "io_direct: -1 (automatic), 0 .. 4096 workgroups"); ctx->opt_io_direct = (int)value; }
    else if (n == "rcd_rows") { if (value != 4 && value != 8) return fail(ctx, ARTGPU_EINVAL, "rcd_rows: 4 or 8"); ctx->opt_rcd_rows = (int)value; }
    else return fail(ctx, ARTGPU_EINVAL, "set_option: unknown option '%s'", name);
    return ARTGPU_OK;
}

int artgpu_set_progress_callback(artgpu_ctx *ctx, artgpu_progress_fn fn, void *user)
{
    if (!ctx) return ARTGPU_EINVAL;
    ctx->progress_fn = fn;
    ctx->progress_user = user;
    return ARTGPU_OK;
}

int artgpu_enable_timing(artgpu_ctx *ctx, int enable)
{
    if (!ctx) return ARTGPU_EINVAL;
    ctx->timing = enable != 0;
    return ARTGPU_OK;
}

int artgpu_get_timings(const artgpu_ctx *ctx, artgpu_timings *out)
{
    if (!ctx || !out) return ARTGPU_EINVAL;
    *out = ctx->last;
    return ARTGPU_OK;
}

size_t artgpu_scratch_bytes(const artgpu_ctx *ctx)
{
    if (!ctx) return 0;
    size_t s = ctx->arena_bytes;
    for (int k = 0; k < artgpu_ctx::NSTAGE; ++k) s += ctx->stage_bytes[k];
    s += ctx->lut_bytes;
    for (int k = 0; k < artgpu_ctx::NPOOL; ++k) s += ctx->pool_bytes[k];
    return s;
}

int artgpu_trim_scratch(artgpu_ctx *ctx)
{
    if (!ctx) return ARTGPU_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // nothing may still be running on what is about to be freed: the context's stream and the side streams it forks to
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->aux) HIPCHK(ctx, hipStreamSynchronize(ctx->aux));
    if (ctx->dn_stream[0]) HIPCHK(ctx, hipStreamSynchronize(ctx->dn_stream[0]));
    if (ctx->amz_side) HIPCHK(ctx, hipStreamSynchronize(ctx->amz_side));
    if (ctx->io_up) HIPCHK(ctx, hipStreamSynchronize(ctx->io_up));
    if (ctx->io_down) HIPCHK(ctx, hipStreamSynchronize(ctx->io_down));
    auto drop = [](float **p, size_t *b) { if (*p) (void)hipFree(*p); *p = nullptr; *b = 0; };
    drop(&ctx->arena, &ctx->arena_bytes);
    for (int k = 0; k < artgpu_ctx::NSTAGE; ++k) drop(&ctx->stage[k], &ctx->stage_bytes[k]);
    for (int k = 0; k < artgpu_ctx::NPOOL; ++k) drop(&ctx->pool[k], &ctx->pool_bytes[k]);
    ctx->batch_px = 0;
    // host-side records of what pool slots held: the tables are gone with them
    ctx->gam_tab = nullptr;
    ctx->ncurve_host.clear();
    for (int k = 0; k < 3; ++k) ctx->rgbcurve_host[k].clear();
    int rc = ARTGPU_OK;
    for (artgpu_ctx *l : ctx->lanes) { const int r = artgpu_trim_scratch(l); if (r && !rc) rc = r; }
    return rc;
}

// Host-side milestones of an entry point: the reference's ProgressListener (rtengine.h:165; amaze_demosaic_RT.cc:1567-1580 reports
// per-tile fractions, the device path reports 0 when the stage's work starts being queued and 1 when the entry point returns) and,
// with the "roctx" option, a roctx range named after the reference function so that profiles read like the reference's call tree.
struct StageScope {
    artgpu_ctx *ctx;
    const char *name;
    bool range;
    typedef int (*push_fn)(const char *);
    typedef int (*pop_fn)();
    static push_fn &push() { static push_fn f = nullptr; return f; }
    static pop_fn &pop() { static pop_fn f = nullptr; return f; }
    StageScope(artgpu_ctx *c, const char *n) : ctx(c), name(n), range(false)
    {
        if (!ctx) return;
        if (ctx->opt_roctx) {
            static bool tried = false;
            if (!tried) {
                tried = true;
                void *h = nullptr;
                for (const char *lib : {"libroctx64.so.4", "libroctx64.so", "/opt/rocm/lib/libroctx64.so"})
                    if ((h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL))) break;
                if (h) {
                    push() = reinterpret_cast<push_fn>(dlsym(h, "roctxRangePushA"));
                    pop() = reinterpret_cast<pop_fn>(dlsym(h, "roctxRangePop"));
                }
            }
            if (push() && pop()) { push()(name); range = true; }
        }
        if (ctx->progress_fn) ctx->progress_fn(ctx->progress_user, name, 0.0);
    }
    ~StageScope()
    {
        if (!ctx) return;
        if (ctx->progress_fn) ctx->progress_fn(ctx->progress_user, name, 1.0);
        if (range) pop()();
    }
};

struct DualReq { double *contrast; int auto_contrast; int second; };
static int vng4_dev(artgpu_ctx *ctx, const float *raw, size_t raw_stride, float *r, float *g, float *b, size_t out_stride, int W, int H, uint32_t filters);
static int dual_blend_dev(artgpu_ctx *ctx, const DevImage &d, int W, int H, uint32_t filters, DualReq *req);
static int demosaic_bayer_impl(artgpu_ctx *ctx, int method, const artgpu_plane *raw, uint32_t filters,
                               double initial_gain, int border, artgpu_rgb *out, DualReq *dual)
{
    StageScope scope_(ctx, "RawImageSource::demosaic (Bayer)");
    if (!ctx) return ARTGPU_EINVAL;
    if (!plane_ok(raw) || !out) return fail(ctx, ARTGPU_EINVAL, "demosaic_bayer: bad raw plane or null output");
    if (method != ARTGPU_BAYER_AMAZE && method != ARTGPU_BAYER_RCD && method != ARTGPU_BAYER_VNG4) return fail(ctx, ARTGPU_EUNSUPPORTED, "demosaic_bayer: method %d is not on the device path", method);
    if (!(initial_gain > 0.0)) return fail(ctx, ARTGPU_EINVAL, "demosaic_bayer: initial_gain must be > 0");
    // RGB Bayer only: the reference falls back to igv_interpolate for 4-colour CFAs (rcd_demosaic.cc:57-66)
    for (unsigned r = 0; r < 2; ++r)
        for (unsigned c = 0; c < 2; ++c)
            if (((filters >> ((((r << 1) & 14) + (c & 1)) << 1)) & 3) == 3) return fail(ctx, ARTGPU_EUNSUPPORTED, "demosaic_bayer: 4-colour CFA");
    const int W = raw->w, H = raw->h;
    if (W < 64 || H < 64) return fail(ctx, ARTGPU_EUNSUPPORTED, "demosaic_bayer: image %dx%d smaller than 64x64", W, H);
    HIPCHK(ctx, hipSetDevice(ctx->device));

    DevImage d;
    int rc = bind_images(ctx, raw, out, &d);
    if (rc) return rc;
    if (ctx->timing) HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream));

    int bord = 0;
    if (method == ARTGPU_BAYER_AMAZE) {
        const int nty = (H + 16 + AMAZE_STEP - 1) / AMAZE_STEP, ntx = (W + 16 + AMAZE_STEP - 1) / AMAZE_STEP;
        const int ntiles = nty * ntx;
        // Tile classes (amaze_demosaic_RT.cc:182-334).  Tiles that write no pixel (the clipped tile is at most 32 wide / high) are
        // skipped.  Tiles that are 160 columns wide go to the LDS streaming kernel (amaze_stream.hip) unless their mirrored bottom /
        // right border fill over-runs the tile row or the cfa plane in the reference (it always writes 16 rows / columns from the
        // frame edge: exact only when a full-size tile ends 16 pixels past the frame); everything else -- narrower tiles, those
        // over-run tiles and streamed tiles whose Nyquist sites do not fit the stream's assumption -- is done by the arena kernel
        // (amaze.hip), which reproduces the reference's buffer layout literally.
        const int mode = ctx->opt_amaze_path;
        if (ctx->amz_w != W || ctx->amz_h != H || ctx->amz_mode != mode) {
            std::vector<int> st, ar;
            for (int ty = 0; ty < nty; ++ty)
                for (int tx = 0; tx < ntx; ++tx) {
                    const int top = -16 + ty * AMAZE_STEP, left = -16 + tx * AMAZE_STEP;
                    const int rr1 = std::min(top + AMAZE_TS, H + 16) - top, cc1 = std::min(left + AMAZE_TS, W + 16) - left;
                    if (rr1 <= 32 || cc1 <= 32) continue;
                    // streamable: 160 columns wide (any height), and the 16 mirrored rows / columns the reference appends at the
                    // bottom / right frame edge end exactly at the tile's edge
                    const bool wide = cc1 == AMAZE_TS && (left + AMAZE_TS <= W || left + AMAZE_TS == W + 16);
                    const bool rows_ok = rr1 < AMAZE_TS || top + AMAZE_TS <= H || top + AMAZE_TS == H + 16;
                    (mode == 0 && wide && rows_ok ? st : ar).push_back(ty * ntx + tx);
                }
            // ... + the redo queue: 4 header ints (reserved, taken, pad) and one 64-bit word per streamed tile, 16-byte aligned
            const size_t nfixed = ((st.size() + (1 + ar.size()) + (1 + ar.size() + st.size()) + 3) / 4) * 4;
            const size_t nints = nfixed + 4 + 2 * st.size() + 8;
            if ((rc = ensure(ctx, &ctx->amz_lists, &ctx->amz_lists_bytes, nints * sizeof(int)))) return rc;
            std::vector<int> hostbuf;
            hostbuf.insert(hostbuf.end(), st.begin(), st.end());
            hostbuf.push_back((int)ar.size());
            hostbuf.insert(hostbuf.end(), ar.begin(), ar.end());
            HIPCHK(ctx, hipMemcpyAsync(ctx->amz_lists, hostbuf.data(), hostbuf.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // hostbuf goes out of scope
            ctx->amz_w = W; ctx->amz_h = H; ctx->amz_mode = mode;
            ctx->amz_nstream = (int)st.size(); ctx->amz_narena = (int)ar.size();
        }
        int *const d_stream = reinterpret_cast<int *>(ctx->amz_lists);
        int *const d_templ = d_stream + ctx->amz_nstream;
        int *const d_work = d_templ + 1 + ctx->amz_narena;
        const size_t nfixed = (((size_t)ctx->amz_nstream + (1 + ctx->amz_narena) + (1 + ctx->amz_narena + ctx->amz_nstream) + 3) / 4) * 4;
        int *const d_queue = d_stream + nfixed;                                          // header (4 ints), then the 64-bit entries
        unsigned long long *const d_qwords = reinterpret_cast<unsigned long long *>(d_queue + 4);
        const float clip_pt = (float)(1.0 / initial_gain);   // amaze_demosaic_RT.cc:53-54
        const float clip_pt8 = (float)(0.8 / initial_gain);
        // The arena kernel runs twice.  EARLY, beside the stream kernel on a stream of its own: the tiles the stream cannot take (for a
        // frame width that is not 32 + a multiple of 128 that is a whole column of narrow tiles: 43 of 2860 at 8256 x 5504) -- they are
        // one workgroup each and a chain of twenty HBM round trips, 0.45 ms that used to follow the stream kernel; now they hold a few CUs
        // back for that long while the stream workgroups on the other CUs take tiles from the shared counter.  LATE, behind both: whatever
        // the stream handed back (tiles whose Nyquist sites did not fit its assumption and found no taker in the redo queue), normally nothing.
        const bool split = mode == 1 && ctx->opt_amaze_split;
        const bool early = ctx->amz_nstream > 0 && ctx->amz_narena > 0 && ctx->opt_amaze_overlap;
        const int nlist_max = ctx->amz_narena + ctx->amz_nstream;
        const int cap = split ? MAX_TILE_WORKGROUPS : 768;
        // (the per-phase profiling path runs one arena per tile of the grid, the tiles that write nothing included)
        const int grid = split ? std::min(ntiles, cap) : std::max(1, std::min(nlist_max, cap));
        rc = ensure(ctx, &ctx->arena, &ctx->arena_bytes, (size_t)grid * AMAZE_ARENA_FLOATS * sizeof(float));
        if (rc) return rc;
        if ((rc = ensure(ctx, &ctx->bbox, &ctx->bbox_bytes, (size_t)grid * 4 * sizeof(int)))) return rc;
        AmazeArgs a;
        a.raw = d.raw; a.raw_stride = d.raw_stride;
        a.red = d.r; a.green = d.g; a.blue = d.b; a.out_stride = d.out_stride;
        a.arena = ctx->arena;
        a.W = W; a.H = H; a.ntx = ntx; a.ntiles = ntiles;
        a.filters = filters;
        a.clip_pt = clip_pt;
        a.clip_pt8 = clip_pt8;
        a.bbox = reinterpret_cast<int *>(ctx->bbox);
        // Arena regions that have read-before-write positions on full tiles and therefore must be cleared per tile: vcd, hcd,
        // vcdalt, hcdalt, cddiffsq, nyquist (bits 4-8, 15): on a full tile every other position is written before it is read, for
        // every CFA phase, because the phase loops cover fixed index ranges (derivation: DESIGN.md section 10).
        // tests/test_gpu_demosaic.py re-checks it with poisoned arenas (artgpu_set_option "amaze_poison").  Partial tiles clear everything.
        a.zero_mask = (unsigned)ctx->opt_amaze_zero_mask;
        // ... and of the five full-size planes among them only positions within a few pixels of the tile edge (plus the gap behind each
        // plane): a frame of 4 already passes the poison test, 16 (the discarded tile border) is used.  0: whole planes.
        a.zero_frame = ctx->opt_amaze_zero_frame;
        a.split = split ? 1 : 0;
        a.queue_words = d_qwords;
        a.queue_counters = reinterpret_cast<int *>(d_qwords + ctx->amz_nstream);
        if (ctx->opt_amaze_poison >= 0)   // test hook: fill the arenas with a byte pattern first
            HIPCHK(ctx, hipMemsetAsync(ctx->arena, ctx->opt_amaze_poison, (size_t)grid * AMAZE_ARENA_FLOATS * sizeof(float), ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(d_queue, 0, (4 + 2 * (size_t)ctx->amz_nstream + 8) * sizeof(int), ctx->stream));   // + the 8 counters behind it
        SideStreamJoin amz_join;
        if (early) {
            // the working list starts empty (the stream appends to it), the static tiles are taken from the template
            HIPCHK(ctx, hipMemsetAsync(d_work, 0, sizeof(int), ctx->stream));
            if (!ctx->amz_side) {
                HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->amz_side, hipStreamNonBlocking));
                for (int k = 0; k < 2; ++k) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->amz_ev[k], hipEventDisableTiming));
            }
            // the STREAM kernel goes to the side stream, behind an event: the arena kernel is then the one the hardware starts first (the
            // other way round the persistent stream workgroups took every CU and the arena tiles waited for them to finish)
            HIPCHK(ctx, hipEventRecord(ctx->amz_ev[0], ctx->stream));          // the CFA plane, the cleared queue
            HIPCHK(ctx, hipStreamWaitEvent(ctx->amz_side, ctx->amz_ev[0], 0));
            amz_join.arm(ctx->amz_side);
            a.tile_list = d_templ + 1; a.tile_count = d_templ;
            a.queue_hdr = nullptr;
            HIPCHK(ctx, launch_amaze(a, std::min(ctx->amz_narena, cap), ctx->stream));
        } else {
            HIPCHK(ctx, hipMemcpyAsync(d_work, d_templ, (size_t)(1 + ctx->amz_narena) * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
        }
        if (ctx->amz_nstream > 0) {
            AmazeStreamArgs sa;
            sa.raw = d.raw; sa.raw_stride = d.raw_stride;
            sa.red = d.r; sa.green = d.g; sa.blue = d.b; sa.out_stride = d.out_stride;
            sa.W = W; sa.H = H; sa.ntx = ntx;
            sa.filters = filters;
            sa.clip_pt = clip_pt; sa.clip_pt8 = clip_pt8;
            const unsigned f00 = (filters >> 0) & 3, f01 = (filters >> 2) & 3;   // FC(0,0), FC(0,1)
            sa.g00 = (int)(f00 & 1);
            sa.ey = f00 == 1 ? (f01 == 0 ? 0 : 1) : (f00 == 0 ? 0 : 1);        // row of the red sites (L1381-1386: ey)
            sa.tiles = d_stream; sa.ntiles = ctx->amz_nstream;
            sa.fallback = d_work;
            sa.queue_hdr = d_queue; sa.queue_words = d_qwords;
            // persistent workgroups, one per CU (the kernel needs almost all of a CU's LDS): each streams tiles back to back
            if (ctx->num_cus <= 0) {
                hipDeviceProp_t prop;
                HIPCHK(ctx, hipGetDeviceProperties(&prop, ctx->device));
                ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            }
            // One stream workgroup owns a CU (156 KB of LDS, sixteen waves at 122 registers) and the kernel is bound by instruction issue; the
            // FTblockDN passes of ANOTHER frame in flight are bound by memory and can do nothing with a CU the demosaic holds.  With frames in
            // flight (artgpu_batch_run lanes) the stream kernel therefore leaves 3/8 of the CUs to them: measured 3881 -> 4006 .. 4056 MP/s with two
            // frames in flight at 45 MP (scripts/r5_lanes.sh: 240 / 224 / 192 / 160 / 144 / 128 workgroups; one frame at a time: all CUs).
            // Option "amaze_grid" overrides.
            const int cu_auto = ctx->frames_in_flight > 1 ? std::max(1, ctx->num_cus * 5 / 8) : ctx->num_cus;
            const int cu_cap = ctx->opt_amaze_grid > 0 ? std::min(ctx->opt_amaze_grid, ctx->num_cus) : cu_auto;
            HIPCHK(ctx, launch_amaze_stream(sa, std::min(ctx->amz_nstream, cu_cap), early ? ctx->amz_side : ctx->stream));
        }
        if (early) {
            HIPCHK(ctx, hipEventRecord(ctx->amz_ev[1], ctx->amz_side));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->amz_ev[1], 0));    // the streamed tiles are written, the hand-back list is final
            amz_join.disarm();
        }
        // arena kernel over the listed tiles (the static ones unless they went early, plus whatever the stream handed back; the count is
        // read on the device)
        if (split) {               // one arena per tile, tiles 0..ntiles-1 (the empty ones write nothing)
            a.tile_list = nullptr; a.tile_count = nullptr;
            if (grid < ntiles) return fail(ctx, ARTGPU_EUNSUPPORTED, "amaze_split needs one arena per tile (%d tiles)", ntiles);
        } else {
            a.tile_list = d_work + 1; a.tile_count = d_work;
        }
        a.queue_hdr = (!split && ctx->amz_nstream > 0) ? d_queue : nullptr;
        HIPCHK(ctx, launch_amaze(a, split ? ntiles : grid, ctx->stream));
        bord = border < 4 ? 3 : 0; // amaze_demosaic_RT.cc:1587-1589
    } else if (method == ARTGPU_BAYER_VNG4) {
        if ((rc = vng4_dev(ctx, d.raw, d.raw_stride, d.r, d.g, d.b, d.out_stride, W, H, filters))) return rc;
        bord = 3; // vng4_demosaic_RT.cc:384
    } else {
        const int tileSizeN = RCD_TS - 2 * RCD_BORDER;      // the reference's tile grid (rcd_demosaic.cc:82-87)
        const int numTh = H / tileSizeN + ((H % tileSizeN) ? 1 : 0), numTw = W / tileSizeN + ((W % tileSizeN) ? 1 : 0);
        const int ntiles = numTh * numTw;
        {
            // persistent workgroups take tiles from a counter; as many as the CUs hold at once
            if (ctx->num_cus <= 0) {
                hipDeviceProp_t prop;
                HIPCHK(ctx, hipGetDeviceProperties(&prop, ctx->device));
                ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            }
            if (!ctx->rcd_counter) HIPCHK(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->rcd_counter), 64));
            HIPCHK(ctx, hipMemsetAsync(ctx->rcd_counter, 0, sizeof(int), ctx->stream));
            RcdStreamArgs a;
            a.raw = d.raw; a.raw_stride = d.raw_stride;
            a.red = d.r; a.green = d.g; a.blue = d.b; a.out_stride = d.out_stride;
            a.W = W; a.H = H; a.numTw = numTw; a.ntiles = ntiles;
            a.filters = filters;
            a.vec2 = d.out_stride % 2 == 0 && (reinterpret_cast<uintptr_t>(d.r) | reinterpret_cast<uintptr_t>(d.g) | reinterpret_cast<uintptr_t>(d.b)) % 8 == 0;
            a.counter = ctx->rcd_counter;
            const int R = ctx->opt_rcd_rows;
            HIPCHK(ctx, launch_rcd_stream(a, R, std::min(ntiles, ctx->num_cus * rcd_stream_workgroups_per_cu(R)), ctx->stream));
        }
        bord = RCD_BORDER; // rcd_demosaic.cc:342
    }
    if (ctx->timing) HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    rc = launch_border(ctx, d, W, H, filters, bord);
    if (rc) return rc;
    if (dual && (rc = dual_blend_dev(ctx, d, W, H, filters, dual))) return rc;
    if (ctx->timing) HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    rc = unbind_images(ctx, out, &d);
    if (rc) return rc;
    if (ctx->timing) {
        HIPCHK(ctx, hipEventSynchronize(ctx->ev[2]));
        HIPCHK(ctx, hipEventElapsedTime(&ctx->last.demosaic_ms, ctx->ev[0], ctx->ev[1]));
        HIPCHK(ctx, hipEventElapsedTime(&ctx->last.border_ms, ctx->ev[1], ctx->ev[2]));
        HIPCHK(ctx, hipEventElapsedTime(&ctx->last.total_ms, ctx->ev[0], ctx->ev[2]));
    }
    return ARTGPU_OK;
}
int artgpu_demosaic_bayer(artgpu_ctx *ctx, int method, const artgpu_plane *raw, uint32_t filters,
                          double initial_gain, int border, artgpu_rgb *out)
{
    return demosaic_bayer_impl(ctx, method, raw, filters, initial_gain, border, out, nullptr);
}
int artgpu_dual_demosaic_bayer(artgpu_ctx *ctx, int method, int second, const artgpu_plane *raw, uint32_t filters, double initial_gain, int border,
                               double *contrast, int auto_contrast, artgpu_rgb *out)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (method == ARTGPU_BAYER_VNG4 || (second != ARTGPU_DUAL_BILINEAR && second != ARTGPU_DUAL_VNG4))
        return fail(ctx, ARTGPU_EUNSUPPORTED, "dual_demosaic_bayer: first demosaicer AMAZE or RCD, second BILINEAR or VNG4");
    if (!contrast || !(*contrast >= 0.0)) return fail(ctx, ARTGPU_EINVAL, "dual_demosaic_bayer: contrast must be >= 0");
    if (raw && (raw->w < 96 || raw->h < 96)) return fail(ctx, ARTGPU_EUNSUPPORTED, "dual_demosaic_bayer: image smaller than 96x96");
    DualReq req = {contrast, auto_contrast, second};
    // contrast == 0 without the automatic threshold: only the first demosaicer runs (dual_demosaic_RT.cc:43-71)
    return demosaic_bayer_impl(ctx, method, raw, filters, initial_gain, border, out, (*contrast == 0.0 && !auto_contrast) ? nullptr : &req);
}

int artgpu_border_interpolate2(artgpu_ctx *ctx, const artgpu_plane *raw, uint32_t filters, int lborders, artgpu_rgb *out)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!plane_ok(raw) || !out || lborders < 0) return fail(ctx, ARTGPU_EINVAL, "border_interpolate2: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevImage d;
    int rc = bind_images(ctx, raw, out, &d);
    if (rc) return rc;
    if (d.staged) {
        // host outputs: start from the caller's current contents so that only the frame changes
        const int W = raw->w, H = raw->h;
        artgpu_plane *op[3] = {&out->r, &out->g, &out->b};
        float *dst[3] = {d.r, d.g, d.b};
        for (int k = 0; k < 3; ++k)
            HIPCHK(ctx, hipMemcpy2DAsync(dst[k], (size_t)W * 4, op[k]->p, (size_t)op[k]->row_stride_bytes, (size_t)W * 4, H, hipMemcpyHostToDevice, ctx->stream));
    }
    rc = launch_border(ctx, d, raw->w, raw->h, filters, lborders);
    if (rc) return rc;
    return unbind_images(ctx, out, &d);
}

int artgpu_get_image(artgpu_ctx *ctx, const artgpu_rgb *planes, int sx1, int sy1, const float mul[3],
                     int do_clip, const double *mat, artgpu_rgb *image)
{
    StageScope scope_(ctx, "RawImageSource::getImage");
    return artgpu_get_image_skip(ctx, planes, sx1, sy1, 1, mul, do_clip, mat, image);
}

int artgpu_get_image_skip(artgpu_ctx *ctx, const artgpu_rgb *planes, int sx1, int sy1, int skip, const float mul[3],
                          int do_clip, const double *mat, artgpu_rgb *image)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!planes || !image || !mul) return fail(ctx, ARTGPU_EINVAL, "get_image: null argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevRGB src, dst;
    int rc = bind_rgb(ctx, planes, 1, true, &src, "get_image(planes)");
    if (rc) return rc;
    rc = bind_rgb(ctx, image, 4, false, &dst, "get_image(image)");
    if (rc) return rc;
    if (skip < 1 || skip > src.w || skip > src.h) return fail(ctx, ARTGPU_EINVAL, "get_image: skip %d", skip);
    // transformRect (rawimagesource.cc:745-747): the caller's image is ceil(crop / skip); the last window may be pulled back inside
    if (sx1 < 0 || sy1 < 0 || sx1 + (dst.w - 1) * skip >= src.w || sy1 + (dst.h - 1) * skip >= src.h)
        return fail(ctx, ARTGPU_EINVAL, "get_image: crop %dx%d+%d+%d (skip %d) outside the %dx%d planes", dst.w, dst.h, sx1, sy1, skip, src.w, src.h);
    PixArgs a = {};
    for (int k = 0; k < 3; ++k) { a.src[k] = src.p[k]; a.dst[k] = dst.p[k]; a.mul[k] = mul[k]; }
    a.src_stride = src.stride; a.dst_stride = dst.stride;
    a.sx1 = sx1; a.sy1 = sy1; a.w = dst.w; a.h = dst.h;
    a.skip = skip; a.src_w = src.w; a.src_h = src.h;
    a.has_mul = 1; a.do_clip = do_clip ? 1 : 0;
    a.has_mat = mat ? 1 : 0;
    if (mat) for (int k = 0; k < 9; ++k) a.mat[k] = mat[k];
    HIPCHK(ctx, launch_get_image_convert(a, ctx->stream));
    return unbind_rgb(ctx, image, &dst);
}

int artgpu_convert_color_space(artgpu_ctx *ctx, artgpu_rgb *image, const double mat[9])
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!image || !mat) return fail(ctx, ARTGPU_EINVAL, "convert_color_space: null argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevRGB d;
    int rc = bind_rgb(ctx, image, 4, true, &d, "convert_color_space");
    if (rc) return rc;
    PixArgs a = {};
    for (int k = 0; k < 3; ++k) { a.src[k] = d.p[k]; a.dst[k] = d.p[k]; }
    a.src_stride = d.stride; a.dst_stride = d.stride; a.w = d.w; a.h = d.h;
    a.has_mul = 0; a.has_mat = 1;
    for (int k = 0; k < 9; ++k) a.mat[k] = mat[k];
    HIPCHK(ctx, launch_get_image_convert(a, ctx->stream));
    return unbind_rgb(ctx, image, &d);
}

int artgpu_exposure(artgpu_ctx *ctx, artgpu_rgb *image, float exp_scale, float black)
{
    StageScope scope_(ctx, "ImProcFunctions::expcomp");
    if (!ctx) return ARTGPU_EINVAL;
    if (!image) return fail(ctx, ARTGPU_EINVAL, "exposure: null image");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevRGB d;
    int rc = bind_rgb(ctx, image, 4, true, &d, "exposure");
    if (rc) return rc;
    PixArgs a = {};
    for (int k = 0; k < 3; ++k) a.dst[k] = d.p[k];
    a.dst_stride = d.stride; a.w = d.w; a.h = d.h;
    a.exp_scale = exp_scale; a.black = black;
    HIPCHK(ctx, launch_exposure(a, ctx->stream));
    return unbind_rgb(ctx, image, &d);
}

// the 65536-entry curve of a tone-curve call -> ctx->lut.  The same curve frame after frame (a batch) is uploaded once: a copy from
// pageable host memory stalls the stream and the enqueueing thread for ~60 us each time.
static int upload_curve(artgpu_ctx *ctx, const float *lut65536)
{
    int rc = ensure(ctx, &ctx->lut, &ctx->lut_bytes, 65536 * sizeof(float));
    if (rc) return rc;
    if (ctx->lut_host.size() == 65536 && std::memcmp(ctx->lut_host.data(), lut65536, 65536 * sizeof(float)) == 0) return ARTGPU_OK;
    ctx->lut_host.assign(lut65536, lut65536 + 65536);
    // from the context's own copy: the caller's array may change as soon as this call returns
    hipError_t e = hipMemcpyAsync(ctx->lut, ctx->lut_host.data(), 65536 * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        ctx->lut_host.clear();      // the cache only describes ctx->lut after a copy that succeeded
        return fail(ctx, ARTGPU_EHIP, "tone curve upload failed: %s", hipGetErrorString(e));
    }
    return ARTGPU_OK;
}

int artgpu_tone_curve(artgpu_ctx *ctx, artgpu_rgb *image, int mode, const float *lut65536, float whitept, int filmlike_clip)
{
    StageScope scope_(ctx, "ImProcFunctions::toneCurve");
    if (!ctx) return ARTGPU_EINVAL;
    if (!image) return fail(ctx, ARTGPU_EINVAL, "tone_curve: null image");
    if (mode != ARTGPU_TONE_STD) return fail(ctx, ARTGPU_EUNSUPPORTED, "tone_curve: curve mode %d is not on the device path", mode);
    if (!(whitept > 0.f)) return fail(ctx, ARTGPU_EINVAL, "tone_curve: whitept %g", (double)whitept);
    if (whitept > 1.f && ctx->curve_tail_kind == ARTGPU_CURVE_TAIL_HOST)
        return fail(ctx, ARTGPU_EUNSUPPORTED, "tone_curve: whitept %g needs the curve beyond the LUT (artgpu_set_curve_tail)", (double)whitept);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevRGB d;
    int rc = bind_rgb(ctx, image, 4, true, &d, "tone_curve");
    if (rc) return rc;
    PixArgs a = {};
    for (int k = 0; k < 3; ++k) a.dst[k] = d.p[k];
    a.dst_stride = d.stride; a.w = d.w; a.h = d.h;
    a.do_clip = filmlike_clip ? 1 : 0; a.whitept = whitept;
    a.tail_kind = ctx->curve_tail_kind == ARTGPU_CURVE_TAIL_HOST ? 0 : ctx->curve_tail_kind; a.tail_y = ctx->curve_tail_y; a.tail_pc = ctx->curve_tail_pc;
    if (lut65536) {
        if ((rc = upload_curve(ctx, lut65536))) return rc;
        a.lut = ctx->lut;
    }
    a.no_lds_lut = !ctx->opt_lut_lds; a.cu_reserve = ctx->cu_reserve;
    HIPCHK(ctx, launch_tone_std(a, ctx->stream));
    return unbind_rgb(ctx, image, &d);
}

// ---------------------------------------------------------------------------------------------
// wavelet_decomposition
// ---------------------------------------------------------------------------------------------
} // extern "C"

struct artgpu_wavelet {
    int w = 0, h = 0, w2 = 0, h2 = 0, nlevels = 0;
    float *bands = nullptr;   // nlevels * 3 * n floats
    float *lowpass[2] = {nullptr, nullptr};
    int cur = 0;              // lowpass[cur] is coeff0
    size_t n = 0;
    float *band(int l, int dir) const { return bands + ((size_t)l * 3 + (dir - 1)) * n; }
};

namespace {
int wavelet_skip(int level) { return level <= 1 ? 1 : 1 << (level - 1); }
}

extern "C" {

int artgpu_wavelet_decompose(artgpu_ctx *ctx, const artgpu_plane *src, int maxlvl, artgpu_wavelet **out)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!plane_ok(src) || !out || maxlvl < 1 || maxlvl > 9) return fail(ctx, ARTGPU_EINVAL, "wavelet_decompose: bad arguments");
    *out = nullptr;
    const int w = src->w, h = src->h, w2 = (w + 1) / 2, h2 = (h + 1) / 2;
    if ((w2 < h2 ? w2 : h2) < 2 * wavelet_skip(maxlvl - 1) || w < 8 || h < 8)
        return fail(ctx, ARTGPU_EUNSUPPORTED, "wavelet_decompose: %dx%d too small for %d levels", w, h, maxlvl);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const float *dsrc = src->p;
    size_t sstride = (size_t)(src->row_stride_bytes / 4);
    if (!src->on_device) {
        int rc = ensure(ctx, &ctx->stage[0], &ctx->stage_bytes[0], (size_t)w * h * 4);
        if (rc) return rc;
        HIPCHK(ctx, hipMemcpy2DAsync(ctx->stage[0], (size_t)w * 4, src->p, (size_t)src->row_stride_bytes, (size_t)w * 4, h, hipMemcpyHostToDevice, ctx->stream));
        dsrc = ctx->stage[0];
        sstride = w;
    }
    artgpu_wavelet *wv = new (std::nothrow) artgpu_wavelet;
    if (!wv) return fail(ctx, ARTGPU_ENOMEM, "wavelet_decompose: out of host memory");
    wv->w = w; wv->h = h; wv->w2 = w2; wv->h2 = h2; wv->nlevels = maxlvl; wv->n = (size_t)w2 * h2;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&wv->bands), (size_t)maxlvl * 3 * wv->n * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&wv->lowpass[0]), wv->n * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&wv->lowpass[1]), wv->n * sizeof(float));
    if (e != hipSuccess) {
        artgpu_wavelet_free(ctx, wv);
        return fail(ctx, ARTGPU_ENOMEM, "wavelet_decompose: hipMalloc failed: %s", hipGetErrorString(e));
    }
    WaveArgs a = {};
    a.w = w; a.h = h; a.w2 = w2; a.h2 = h2;
    for (int l = 0; l < maxlvl; ++l) {
        a.b1 = wv->band(l, 1); a.b2 = wv->band(l, 2); a.b3 = wv->band(l, 3);
        hipError_t le;
        if (l == 0) {
            a.src = dsrc; a.src_stride = sstride; a.lo = wv->lowpass[0];
            wv->cur = 0;
            le = launch_wavelet_analysis0(a, ctx->stream);
        } else {
            a.src = wv->lowpass[wv->cur]; a.lo = wv->lowpass[wv->cur ^ 1]; a.skip = wavelet_skip(l);
            le = launch_wavelet_haar_analysis(a, ctx->stream);
            wv->cur ^= 1;
        }
        if (le != hipSuccess) {
            artgpu_wavelet_free(ctx, wv);
            return fail(ctx, ARTGPU_EHIP, "wavelet_decompose: launch failed: %s", hipGetErrorString(le));
        }
    }
    if (!src->on_device) {
        const hipError_t se = hipStreamSynchronize(ctx->stream);
        if (se != hipSuccess) {
            artgpu_wavelet_free(ctx, wv);
            return fail(ctx, ARTGPU_EHIP, "wavelet_decompose: %s", hipGetErrorString(se));
        }
    }
    *out = wv;
    return ARTGPU_OK;
}

int artgpu_wavelet_info(const artgpu_wavelet *wv, int32_t *w2, int32_t *h2, int32_t *nlevels)
{
    if (!wv) return ARTGPU_EINVAL;
    if (w2) *w2 = wv->w2;
    if (h2) *h2 = wv->h2;
    if (nlevels) *nlevels = wv->nlevels;
    return ARTGPU_OK;
}

int artgpu_wavelet_get_band(artgpu_ctx *ctx, const artgpu_wavelet *wv, int level, int dir, float *dst, int on_device)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!wv || !dst || dir < 0 || dir > 3 || (dir > 0 && (level < 0 || level >= wv->nlevels))) return fail(ctx, ARTGPU_EINVAL, "wavelet_get_band: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const float *p = dir == 0 ? wv->lowpass[wv->cur] : wv->band(level, dir);
    HIPCHK(ctx, hipMemcpyAsync(dst, p, wv->n * sizeof(float), on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    if (!on_device) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ARTGPU_OK;
}

int artgpu_wavelet_set_band(artgpu_ctx *ctx, artgpu_wavelet *wv, int level, int dir, const float *src, int on_device)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!wv || !src || dir < 0 || dir > 3 || (dir > 0 && (level < 0 || level >= wv->nlevels))) return fail(ctx, ARTGPU_EINVAL, "wavelet_set_band: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    float *p = dir == 0 ? wv->lowpass[wv->cur] : wv->band(level, dir);
    HIPCHK(ctx, hipMemcpyAsync(p, src, wv->n * sizeof(float), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    if (!on_device) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ARTGPU_OK;
}

int artgpu_wavelet_reconstruct(artgpu_ctx *ctx, artgpu_wavelet *wv, artgpu_plane *dst, float blend)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!wv || !plane_ok(dst) || dst->w != wv->w || dst->h != wv->h) return fail(ctx, ARTGPU_EINVAL, "wavelet_reconstruct: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    float *ddst = dst->p;
    size_t dstride = (size_t)(dst->row_stride_bytes / 4);
    if (!dst->on_device) {
        int rc = ensure(ctx, &ctx->stage[0], &ctx->stage_bytes[0], (size_t)wv->w * wv->h * 4);
        if (rc) return rc;
        // dst participates in the blend (dst*(1-blend) + blend*4*tot): bring the host contents over
        HIPCHK(ctx, hipMemcpy2DAsync(ctx->stage[0], (size_t)wv->w * 4, dst->p, (size_t)dst->row_stride_bytes, (size_t)wv->w * 4, wv->h, hipMemcpyHostToDevice, ctx->stream));
        ddst = ctx->stage[0];
        dstride = wv->w;
    }
    WaveArgs a = {};
    a.w = wv->w; a.h = wv->h; a.w2 = wv->w2; a.h2 = wv->h2; a.blend = blend;
    for (int l = wv->nlevels - 1; l > 0; --l) {
        a.src = wv->lowpass[wv->cur]; a.lo = wv->lowpass[wv->cur ^ 1];
        a.b1 = wv->band(l, 1); a.b2 = wv->band(l, 2); a.b3 = wv->band(l, 3); a.skip = wavelet_skip(l);
        HIPCHK(ctx, launch_wavelet_haar_synthesis(a, ctx->stream));
        wv->cur ^= 1;
    }
    a.src = wv->lowpass[wv->cur];
    a.b1 = wv->band(0, 1); a.b2 = wv->band(0, 2); a.b3 = wv->band(0, 3);
    a.dst = ddst; a.dst_stride = dstride;
    HIPCHK(ctx, launch_wavelet_synthesis0(a, ctx->stream));
    if (!dst->on_device) {
        HIPCHK(ctx, hipMemcpy2DAsync(dst->p, (size_t)dst->row_stride_bytes, ddst, (size_t)wv->w * 4, (size_t)wv->w * 4, wv->h, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return ARTGPU_OK;
}

int artgpu_wavelet_free(artgpu_ctx *ctx, artgpu_wavelet *wv)
{
    if (!wv) return ARTGPU_EINVAL;
    if (ctx) { (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->stream); }
    if (wv->bands) (void)hipFree(wv->bands);
    if (wv->lowpass[0]) (void)hipFree(wv->lowpass[0]);
    if (wv->lowpass[1]) (void)hipFree(wv->lowpass[1]);
    delete wv;
    return ARTGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// RGB_denoise (wavelet part)
// ---------------------------------------------------------------------------------------------
} // extern "C"

namespace {

enum { P_L = 0, P_A, P_B, P_LBANDS, P_LLOW0, P_LLOW1, P_CBANDS, P_CLOW0, P_CLOW1, P_SF, P_TMP, P_HISTO, P_MAD, P_GAM, P_CCALC, P_LIN, P_BLOCKS, P_DTAB, P_CCMAP, P_CACHEF, P_PQ, P_XCBRT, P_GAUSS64, P_DMASK, P_LABTABS, P_PIPE_R, P_PIPE_G, P_PIPE_B, P_DNINFO, P_BATCH, P_DCTTAB, P_CBANDS2, P_CLOW0_2, P_CLOW1_2, P_SF_A, P_SF_B, P_HISTO_A, P_HISTO_B, P_RGBCURVES, P_FUSED, P_LBANDS2,
       P_IO_IN0, P_IO_IN1, P_IO_CFA, P_IO_IMG0, P_IO_IMG1, P_IO_IMG2, P_IO_IMG3, P_IO_IMG4, P_IO_IMG5, P_IO_OUT0, P_IO_OUT1, P_IO_FLAGS,      // artgpu_batch_run_io: staging slots, the CFA plane, the working image, the frames' flag words
       P_NSLOTS };
static_assert(P_NSLOTS <= artgpu_ctx::NPOOL, "grow artgpu_ctx::pool");

struct DevDecomp {
    float *bands, *low[2];
    int cur, nlevels, w, h, w2, h2;
    size_t n;
    float *band(int l, int dir) const { return bands + ((size_t)l * 3 + (dir - 1)) * n; }
};

int decompose_dev(artgpu_ctx *ctx, DevDecomp &d, const float *src, hipStream_t st = nullptr)
{
    if (!st) st = ctx->stream;
    WaveArgs a = {};
    a.w = d.w; a.h = d.h; a.w2 = d.w2; a.h2 = d.h2;
    for (int l = 0; l < d.nlevels; ++l) {
        a.b1 = d.band(l, 1); a.b2 = d.band(l, 2); a.b3 = d.band(l, 3);
        if (l == 0) {
            a.src = src; a.src_stride = d.w; a.lo = d.low[0]; d.cur = 0;
            HIPCHK(ctx, launch_wavelet_analysis0(a, st));
        } else {
            a.src = d.low[d.cur]; a.lo = d.low[d.cur ^ 1]; a.skip = wavelet_skip(l);
            HIPCHK(ctx, launch_wavelet_haar_analysis(a, st));
            d.cur ^= 1;
        }
    }
    return ARTGPU_OK;
}

int reconstruct_dev(artgpu_ctx *ctx, DevDecomp &d, float *dst, hipStream_t st = nullptr)
{
    if (!st) st = ctx->stream;
    WaveArgs a = {};
    a.w = d.w; a.h = d.h; a.w2 = d.w2; a.h2 = d.h2; a.blend = 1.f;
    for (int l = d.nlevels - 1; l > 0; --l) {
        a.src = d.low[d.cur]; a.lo = d.low[d.cur ^ 1];
        a.b1 = d.band(l, 1); a.b2 = d.band(l, 2); a.b3 = d.band(l, 3); a.skip = wavelet_skip(l);
        HIPCHK(ctx, launch_wavelet_haar_synthesis(a, st));
        d.cur ^= 1;
    }
    a.src = d.low[d.cur];
    a.b1 = d.band(0, 1); a.b2 = d.band(0, 2); a.b3 = d.band(0, 3);
    a.dst = dst; a.dst_stride = d.w;
    HIPCHK(ctx, launch_wavelet_synthesis0(a, st));
    return ARTGPU_OK;
}

int pool_get(artgpu_ctx *ctx, int slot, size_t bytes, float **out)
{
    int rc = ensure(ctx, &ctx->pool[slot], &ctx->pool_bytes[slot], bytes);
    *out = ctx->pool[slot];
    return rc;
}

} // namespace

extern "C" {

namespace {
int detail_mask_dev(artgpu_ctx *ctx, const float *src, size_t src_stride, float *mask, int W, int H,
                    float scaling, float threshold, float ceiling, float factor, float blur, float *scratch);
}

int artgpu_wavelet_mad(artgpu_ctx *ctx, const artgpu_wavelet *wv, float *mad_sqr)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!wv || !mad_sqr) return fail(ctx, ARTGPU_EINVAL, "wavelet_mad: null argument");
    if (wv->n > 0x7fffffff) return fail(ctx, ARTGPU_EUNSUPPORTED, "wavelet_mad: band larger than an int holds (MadRgb's datalen is an int)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int nsub = 3 * wv->nlevels;
    float *histo_f, *mad;
    int rc;
    if ((rc = pool_get(ctx, P_HISTO, (size_t)nsub * (65536 + MAD_SCRATCH_INTS_PER_BAND) * 4, &histo_f)) || (rc = pool_get(ctx, P_MAD, 3 * 32 * 4, &mad))) return rc;
    HIPCHK(ctx, launch_mad(wv->bands, wv->n, nsub, reinterpret_cast<int *>(histo_f), mad, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(mad_sqr, mad, (size_t)nsub * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ARTGPU_OK;
}

int artgpu_rgb_denoise(artgpu_ctx *ctx, artgpu_rgb *img, const artgpu_denoise_params *p, const float ws[9], const float *iws,
                       double expcomp, double scale, const artgpu_plane *ccalc, uint32_t flags,
                       float *nresi, float *highresi)
{
    StageScope scope_(ctx, "denoise::RGB_denoise");
    if (!ctx) return ARTGPU_EINVAL;
    if (!img || !p || !ws) return fail(ctx, ARTGPU_EINVAL, "rgb_denoise: null argument");
    if (p->color_space != 0 && p->color_space != 1) return fail(ctx, ARTGPU_EINVAL, "rgb_denoise: color_space must be 0 (RGB) or 1 (LAB)");
    const bool lab_mode = p->color_space == 1;
    if (lab_mode && !iws) return fail(ctx, ARTGPU_EINVAL, "rgb_denoise: LAB mode needs the inverse working-space matrix");
    if (p->chrominance_method != 0 && p->chrominance_method != 1) return fail(ctx, ARTGPU_EINVAL, "rgb_denoise: chrominance_method must be 0 (MANUAL) or 1 (AUTOMATIC)");
    const bool do_detail = !(flags & ARTGPU_DN_SKIP_DETAIL_RECOVERY);
    if (!(scale >= 1.0)) return fail(ctx, ARTGPU_EINVAL, "rgb_denoise: scale must be >= 1");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevRGB d;
    int rc = bind_rgb(ctx, img, 4, true, &d, "rgb_denoise");
    if (rc) return rc;
    const int w = d.w, h = d.h, w2 = (w + 1) / 2, h2 = (h + 1) / 2;
    if (w > 32767 || h > 32767) return fail(ctx, ARTGPU_EUNSUPPORTED, "rgb_denoise: the reference holds the size in short (FTblockDN.cc:1779)");
    const size_t n = (size_t)w * h, n2 = (size_t)w2 * h2;
    const bool useNoiseCCurve = ccalc != nullptr;
    if (p->luminance == 0 && p->chrominance == 0 && !useNoiseCCurve) return unbind_rgb(ctx, img, &d); // L1655-1668: nothing to do

    // ---- scalar set-up, as the reference computes it on the host (L1687-1688,1795-1825,2032-2082,2246-2293)
    const float noiseluma = (float)p->luminance;
    const double nl_t = (noiseluma / 125.0) * (1.0 + noiseluma / 25.0);
    const float noisevarL = (float)(nl_t * nl_t);
    const bool denoiseLuminance = noisevarL > 0.00001f;
    const float gam = (float)p->gamma;
    const float gamthresh = 0.001f;
    const float gamslope = (float)(std::exp(std::log((double)gamthresh) / gam) / gamthresh);
    const float igam = 1.f / gam, igamthresh = gamthresh * gamslope, igamslope = 1.f / gamslope;
    const float gain = std::pow(2.0f, float(expcomp));
    const float interm_med = (float)p->chrominance / 10.0;
    float intermred = p->chrominance_red_green > 0. ? (p->chrominance_red_green / 10.) : (float)p->chrominance_red_green / 7.0;
    float intermblue = p->chrominance_blue_yellow > 0. ? (p->chrominance_blue_yellow / 10.) : (float)p->chrominance_blue_yellow / 7.0;
    float realred = interm_med + intermred;
    if (realred <= 0.f) realred = 0.001f;
    float realblue = interm_med + intermblue;
    if (realblue <= 0.f) realblue = 0.001f;
    const float noisevarab_r = realred * realred, noisevarab_b = realblue * realblue;
    const float maxNoiseVarab = noisevarab_b > noisevarab_r ? noisevarab_b : noisevarab_r;
    int levwav = 5;
    const float maxreal = realred > realblue ? realred : realblue;
    if (maxreal < 8.f) levwav = 5; else if (maxreal < 10.f) levwav = 6; else if (maxreal < 15.f) levwav = 7; else levwav = 8;
    const bool aggressive = p->aggressive != 0;          // QUALITY_HIGH (L1671)
    if (aggressive) levwav += 2;                         // L2260-2262
    if (levwav > 8) levwav = 8;
    { const int t = int(levwav - std::ceil(std::log(scale))); levwav = t > 5 ? t : 5; }
    const int minsizetile = w < h ? w : h;
    int maxlev2 = 8;
    if (minsizetile < 256) maxlev2 = 7;
    if (minsizetile < 128) maxlev2 = 6;
    if (minsizetile < 64) maxlev2 = 5;
    if (minsizetile < 32) maxlev2 = 4;
    if (minsizetile < 16) maxlev2 = 3;
    levwav = levwav < maxlev2 ? levwav : maxlev2;
    if ((w2 < h2 ? w2 : h2) < 2 * wavelet_skip(levwav - 1) || w < 8 || h < 8)
        return fail(ctx, ARTGPU_EUNSUPPORTED, "rgb_denoise: %dx%d too small for %d wavelet levels on the device path", w, h, levwav);
    const int nsub = 3 * levwav;
    bool autoch = p->chrominance_method == 1;

    // ---- streams.  The reference runs a, then b, then L (L2328-2438).  The chains only meet in the untouched L coefficients and their MADs
    // (read by the chroma shrink factors) and in yuv2rgb, so their order is free.  With option "dn_streams" 1 the DCT detail recovery of L -- bound by
    // instruction issue, it leaves HBM idle -- runs on a side stream beside the box blurs and reconstructions of a and b, which are bound by
    // HBM: L goes first for that, after the chroma shrink factors have read its coefficients.  Same kernels on the same data: the same bits.
    // (Rounds 3 and 4: -0.3 ms per frame, the default.  Round 5: off by default -- see opt_dn_streams.)
    // (Running all three chains side by side was measured too: 9.7 ms against 9.2 -- three HBM-bound chains only get in each other's way.)
    const bool fork = ctx->opt_dn_streams != 0 && do_detail && denoiseLuminance;

    // ---- device buffers; nothing is allocated in steady state.  Scratch planes are shared wherever the order of the kernels allows it:
    // one `tmp` (horizontally blurred factors) serves all three channels -- its producer and its consumer are neighbours on the context's
    // stream --; in the reference's order one `sf` plane set and one chroma decomposition do too, with the side stream a and b keep theirs
    // across the L chain.
    float *L, *A, *B, *gamlut, *mad, *ccalc_dev = nullptr, *tmp1 = nullptr;
    float *sfc[3] = {nullptr, nullptr, nullptr}, *tmpc[3], *histo_fc[3];          // 0: L, 1: a, 2: b
    DevDecomp Ld = {}, Cdd[2] = {};
    Ld.w = w; Ld.h = h; Ld.w2 = w2; Ld.h2 = h2; Ld.n = n2; Ld.nlevels = levwav;
    Cdd[0] = Cdd[1] = Ld;
    BlurArgs bl0 = {};
    bl0.n = n2; bl0.w = w2; bl0.h = h2;
    int maxrad = 1;
    for (int l = 0; l < levwav; ++l) { const int r = int((l + 2) / scale); bl0.rad[l] = r > 1 ? r : 1; maxrad = bl0.rad[l] > maxrad ? bl0.rad[l] : maxrad; }
    // ShrinkAllL / ShrinkAllAB as one kernel per channel (shrinkblur.hip: factors, both running sums and the coefficient update in one pass
    // over the coefficients) instead of three with the factor and the row-blurred planes in between: no `sf` / `tmp` planes at all
    const bool fused = ctx->opt_dn_fused != 0 && shrink_blur_supported(w2, h2, bl0.rad, 0, nsub);
    // ... and for all three channels in ONE launch where nothing has to happen between them (no residuals to read back, no BiShrink
    // passes, every level of L shrunk, both chroma channels denoised): the strips of a band can only follow each other a few blocks apart, so
    // it takes the bands of all channels to keep every CU busy, and one launch instead of three has one ragged end instead of three
    const bool merged = fused && ctx->opt_dn_fused != 2 && !aggressive && !nresi && !highresi && denoiseLuminance && levwav <= 5 &&
                        (autoch || (noisevarab_r > 0.001f && noisevarab_b > 0.001f));
    const bool merged_mad = merged && ctx->opt_dn_fused != 3;      // (3: test switch, MadRgb per channel)
    const bool two_chroma = fork || merged;        // a and b keep their own decomposition (otherwise b reuses a's)
    float *fused_scratch = nullptr, *Lbands2 = nullptr;
    const size_t histo_bytes = (size_t)nsub * (65536 + MAD_SCRATCH_INTS_PER_BAND) * 4, band_bytes = (size_t)nsub * n2 * 4;
    if ((rc = pool_get(ctx, P_L, n * 4, &L)) || (rc = pool_get(ctx, P_A, n * 4, &A)) || (rc = pool_get(ctx, P_B, n * 4, &B)) ||
        (rc = pool_get(ctx, P_LBANDS, (merged_mad ? 3 : 1) * band_bytes, &Ld.bands)) || (rc = pool_get(ctx, P_LLOW0, n2 * 4, &Ld.low[0])) || (rc = pool_get(ctx, P_LLOW1, n2 * 4, &Ld.low[1])) ||
        (rc = merged_mad ? ARTGPU_OK : pool_get(ctx, P_CBANDS, (merged ? 2 : 1) * band_bytes, &Cdd[0].bands)) || (rc = pool_get(ctx, P_CLOW0, n2 * 4, &Cdd[0].low[0])) || (rc = pool_get(ctx, P_CLOW1, n2 * 4, &Cdd[0].low[1])) ||
        (rc = pool_get(ctx, P_HISTO, (merged_mad ? 3 : 1) * histo_bytes, &histo_fc[0])) ||
        (rc = pool_get(ctx, P_MAD, 3 * 32 * 4, &mad)) || (rc = pool_get(ctx, P_GAM, 2 * 65536 * 4, &gamlut)))
        return rc;
    // a context that changes between the two forms of the passes gives back what only the other form uses (everything else is shared)
    auto pool_drop = [&](int slot) -> int {
        if (!ctx->pool[slot]) return ARTGPU_OK;
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->dn_stream[0]) HIPCHK(ctx, hipStreamSynchronize(ctx->dn_stream[0]));
        HIPCHK(ctx, hipFree(ctx->pool[slot]));
        ctx->pool[slot] = nullptr; ctx->pool_bytes[slot] = 0;
        return ARTGPU_OK;
    };
    // What only the OTHER form of the shrink passes uses (band-sized slots) is given back, but not on every switch: `fused` depends on the
    // frame (w2, h2 >= 64, radii <= 15), so a context or a batch that alternates small and large frames would pay a stream drain and a band-sized
    // hipFree / hipMalloc per frame -- the allocation the pool exists to avoid.  The slots go after FOUR calls in a row in the same form
    // (artgpu_trim_scratch returns everything at once).
    if (ctx->dn_form == (fused ? 1 : 0)) ++ctx->dn_form_streak; else { ctx->dn_form = fused ? 1 : 0; ctx->dn_form_streak = 1; }
    const bool settle = ctx->dn_form_streak >= 4;
    if (fused) {
        if (settle && ((rc = pool_drop(P_SF_A)) || (rc = pool_drop(P_SF_B)) || (merged && (rc = pool_drop(P_CBANDS2))))) return rc;
        if (!ctx->fs_diag) {      // (optional: without it a timed-out wait still traps, merely unattributed)
            if (hipHostMalloc(reinterpret_cast<void **>(&ctx->fs_diag), 64, hipHostMallocDefault) == hipSuccess) std::memset(ctx->fs_diag, 0, 64);
            else { ctx->fs_diag = nullptr; (void)hipGetLastError(); }
        }
        if ((rc = pool_get(ctx, P_FUSED, shrink_blur_scratch_floats(w2, h2, merged ? 3 * nsub : nsub, maxrad) * 4, &fused_scratch))) return rc;
        // the chroma factors need the L coefficients as the decomposition left them: whenever they are evaluated beside or after the L pass
        // (one launch for all channels; the side stream's order) the L pass writes a second band set instead of updating the first
        if ((fork || merged) && (rc = pool_get(ctx, P_LBANDS2, band_bytes, &Lbands2))) return rc;
    } else if ((settle && ((rc = pool_drop(P_FUSED)) || (rc = pool_drop(P_LBANDS2)))) || (rc = pool_get(ctx, P_SF, band_bytes, &sfc[0])) || (rc = pool_get(ctx, P_TMP, band_bytes, &tmp1))) return rc;
    tmpc[0] = tmpc[1] = tmpc[2] = tmp1;
    if (two_chroma) {
        if (merged_mad) Cdd[0].bands = Ld.bands + (size_t)nsub * n2;       // (one MadRgb launch set walks the bands of all three channels)
        if (merged) Cdd[1].bands = Cdd[0].bands + (size_t)nsub * n2;       // (one launch walks both channels' bands: back to back)
        else if ((rc = pool_get(ctx, P_CBANDS2, band_bytes, &Cdd[1].bands))) return rc;
        if ((rc = pool_get(ctx, P_CLOW0_2, n2 * 4, &Cdd[1].low[0])) || (rc = pool_get(ctx, P_CLOW1_2, n2 * 4, &Cdd[1].low[1])) ||
            (rc = pool_get(ctx, P_HISTO_A, (merged ? 2 : 1) * histo_bytes, &histo_fc[1])) || (rc = pool_get(ctx, P_HISTO_B, histo_bytes, &histo_fc[2])))
            return rc;
        if (!fused && ((rc = pool_get(ctx, P_SF_A, band_bytes, &sfc[1])) || (rc = pool_get(ctx, P_SF_B, band_bytes, &sfc[2])))) return rc;
    } else {
        sfc[1] = sfc[2] = sfc[0];
        histo_fc[1] = histo_fc[2] = histo_fc[0];
        Cdd[1] = Cdd[0];
    }
    float *madL = mad;
    if (useNoiseCCurve) {
        if (!plane_ok(ccalc) || ccalc->w != w2 || ccalc->h != h2) return fail(ctx, ARTGPU_EINVAL, "rgb_denoise: ccalc must be %dx%d", w2, h2);
        if (ccalc->on_device && (size_t)ccalc->row_stride_bytes == (size_t)w2 * 4) {
            ccalc_dev = ccalc->p;              // dense and on the device already (the map artgpu_improc_denoise has just computed): read in place
        } else {
            if ((rc = pool_get(ctx, P_CCALC, n2 * 4, &ccalc_dev))) return rc;
            HIPCHK(ctx, hipMemcpy2DAsync(ccalc_dev, (size_t)w2 * 4, ccalc->p, (size_t)ccalc->row_stride_bytes, (size_t)w2 * 4, h2,
                                         ccalc->on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
        }
    }

    if (fork && !ctx->dn_stream[0]) {
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->dn_stream[0], hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->dn_ev[k], hipEventDisableTiming));
    }
    hipStream_t sL = ctx->stream;

    // ---- gamma LUTs (built on the device with the SSE-form sleef, color.cc:1128-1161)
    //      -- once per set of parameters: the pair built by the previous call on this context is still there when they have not changed
    {
        const float key[6] = {gam, gamthresh, gamslope, igam, igamthresh, igamslope};
        if (ctx->gam_tab != gamlut || std::memcmp(ctx->gam_key, key, sizeof key) != 0) {
            ctx->gam_tab = nullptr;
            HIPCHK(ctx, launch_gamma_lut(gamlut, gam, gamthresh, gamslope, 65535.f, 65535.f, sL));
            HIPCHK(ctx, launch_gamma_lut(gamlut + 65536, igam, igamthresh, igamslope, 65535.f, 65535.f, sL));
            std::memcpy(ctx->gam_key, key, sizeof key);
            ctx->gam_tab = gamlut;
        }
    }

    DnPixArgs px = {};
    for (int k = 0; k < 3; ++k) { px.rgb[k] = d.p[k]; px.ws1[k] = ws[3 + k]; }
    px.stride = d.stride; px.L = L; px.A = A; px.B = B; px.w = w; px.h = h;
    px.gain = gain; px.newGain = 1.f / gain;
    px.gam = gam; px.gamthresh = gamthresh; px.gamslope = gamslope; px.igam = igam; px.igamthresh = igamthresh; px.igamslope = igamslope;
    px.gamcurve = gamlut; px.igamcurve = gamlut + 65536;
    px.pre_scale = ctx->fuse_pre; px.post_scale = ctx->fuse_post; px.no_lds_lut = !ctx->opt_lut_lds; px.cu_reserve = ctx->cu_reserve;
    px.gi = ctx->fuse_gi; px.exp_on = ctx->fuse_exp_on; px.exp_scale = ctx->fuse_exp_scale; px.exp_black = ctx->fuse_exp_black;
    // the inverse-gamma pass looks up gamma-encoded values: the mid-tones sit in the middle of the table, so the 40704 entries kept in LDS
    // start at 8000 (gamma 1.7: linear 0.03 .. 0.60 of white); the forward pass and the tone curve index with linear data and keep [0, 40704)
    px.igam_lds_lo = 8000;
    if (lab_mode) {
        // Color::cachef / cachefy / denoiseGammaTab / denoiseIGammaTab, built on the host like the reference's (color.cc:202-292)
        float *tabs;
        const bool fresh = ctx->pool[P_LABTABS] == nullptr;
        if ((rc = pool_get(ctx, P_LABTABS, 4 * 65536 * 4, &tabs))) return rc;
        if (fresh) {
            std::vector<float> host(4 * 65536);
            build_cachef(host.data()); build_cachefy(host.data() + 65536);
            build_denoise_gamma_tabs(host.data() + 2 * 65536, host.data() + 3 * 65536);
            hipError_t e = hipMemcpyAsync(tabs, host.data(), host.size() * 4, hipMemcpyHostToDevice, sL);
            if (e == hipSuccess) e = hipStreamSynchronize(sL);
            if (e != hipSuccess) {     // the slot doubles as the "tables are there" flag: give it back, or the next call would skip the upload
                (void)hipFree(ctx->pool[P_LABTABS]); ctx->pool[P_LABTABS] = nullptr; ctx->pool_bytes[P_LABTABS] = 0;
                return fail(ctx, ARTGPU_EHIP, "rgb_denoise: upload of the Lab tables failed: %s", hipGetErrorString(e));
            }
        }
        px.lab_mode = 1;
        px.cachef = tabs; px.cachefy = tabs + 65536; px.dn_gamma = tabs + 2 * 65536; px.dn_igamma = tabs + 3 * 65536;
        for (int k = 0; k < 9; ++k) { px.wpi[k] = ws[k]; px.iws[k] = iws[k]; }
    }
    px.realred = realred; px.realblue = realblue; px.qhighFactor = aggressive ? 1.f / static_cast<float>(0.9) : 1.0f;   // L1672
    HIPCHK(ctx, launch_rgb2yuv(px, sL));

    // ---- L decomposition and its MADs (L2296-2320)
    if ((rc = decompose_dev(ctx, Ld, L, sL))) return rc;
    if (!merged_mad) HIPCHK(ctx, launch_mad(Ld.bands, n2, nsub, reinterpret_cast<int *>(histo_fc[0]), madL, sL));

    // one fused ShrinkAll pass over `nb` bands starting at level `lev0` (pointers already offset to the first band)
    auto fused_pass = [&](bool ab, const float *cin, float *cout, const float *cL, const float *mL, const float *mab, int lev0, int nb,
                          float nv_const, float noisevar_ab) -> int {
        FusedShrinkArgs fa = {};
        if (ab) { fa.coefC = cout; fa.nL = 0; fa.nsub_ch = nb; }           // (chroma bands are updated in place: cin == cout)
        else { fa.coef = cin; fa.coef_out = cout; fa.nL = nb; }
        fa.coefL = cL; fa.n = n2; fa.w = w2; fa.h = h2;
        fa.madL = mL; fa.madab = mab;
        fa.noisevar = ccalc_dev; fa.noisevar_nonneg = ctx->ccalc_nonneg; fa.noisevar_const = nv_const; fa.noisevar_scale = maxNoiseVarab; fa.noisevar_ab[0] = fa.noisevar_ab[1] = noisevar_ab;
        fa.useNoiseCCurve = useNoiseCCurve ? 1 : 0;
        for (int l = 0; l < 10; ++l) fa.rad[l] = bl0.rad[l];
        fa.level0 = lev0; fa.nsub = nb; fa.diag = ctx->fs_diag; fa.wait_ticks = (long long)ctx->opt_dn_wait_ms * 100000LL; fa.stall_band = ctx->opt_dn_debug_stall < 0 ? -1 : ctx->opt_dn_debug_stall >> 16; fa.stall_strip = ctx->opt_dn_debug_stall < 0 ? -1 : ctx->opt_dn_debug_stall & 0xffff;
        HIPCHK(ctx, launch_shrink_blur(fa, fused_scratch, sL));
        return ARTGPU_OK;
    };

    // ---- a and b (L2328-2402), first half: decompose, MADs, shrink factors against the untouched L coefficients
    float noisevar_abc[2];
    auto chroma_front = [&](int ch) -> int {
        DevDecomp &Cd = Cdd[ch];
        float *sf = sfc[1 + ch], *tmp = tmpc[1 + ch], *madab = mad + 32 * (1 + ch);
        int *histo = reinterpret_cast<int *>(histo_fc[1 + ch]);
        float noisevar_ab = ch == 0 ? noisevarab_r : noisevarab_b;
        if (autoch && noisevar_ab <= 0.001f) noisevar_ab = 0.02f;
        noisevar_abc[ch] = noisevar_ab;
        int rc2;
        if ((rc2 = decompose_dev(ctx, Cd, ch == 0 ? A : B, sL))) return rc2;
        if (aggressive && noisevar_ab > 0.001f) {
            // WaveletDenoiseAll_BiShrinkAB (L976-1108): MAD of all untouched bands, ShrinkAllAB on the top level (same MAD),
            // point-wise shrink of the levels below
            HIPCHK(ctx, launch_mad(Cd.bands, n2, nsub, histo, madab, sL));
            ShrinkArgs sa = {};
            sa.n = n2; sa.noisevar = ccalc_dev; sa.noisevar_scale = maxNoiseVarab; sa.noisevar_ab = noisevar_ab; sa.useNoiseCCurve = useNoiseCCurve ? 1 : 0;
            const size_t top = (size_t)(nsub - 3) * n2;
            sa.coef = Cd.bands + top; sa.coefL = Ld.bands + top; sa.sfave = sf; sa.madL = madL + (nsub - 3); sa.madab = madab + (nsub - 3);
            if (fused) {
                if ((rc2 = fused_pass(true, Cd.bands + top, Cd.bands + top, Ld.bands + top, madL + (nsub - 3), madab + (nsub - 3), levwav - 1, 3, 0.f, noisevar_ab))) return rc2;
            } else {
                HIPCHK(ctx, launch_shrink_sf(sa, 3, true, sL));
                BlurArgs bt = bl0;
                bt.level0 = levwav - 1;
                bt.src = sf; bt.dst = tmp;
                HIPCHK(ctx, launch_hblur(bt, 3, sL));
                bt.src = tmp; bt.sfave = sf; bt.coef = Cd.bands + top;
                HIPCHK(ctx, launch_vblur_combine(bt, 3, sL));
            }
            if (nsub > 3) {
                sa.coef = Cd.bands; sa.coefL = Ld.bands; sa.madL = madL; sa.madab = madab;
                HIPCHK(ctx, launch_bishrink_AB(sa, nsub - 3, sL));
            }
        }
        if (noisevar_ab > 0.001f && !merged_mad) {
            HIPCHK(ctx, launch_mad(Cd.bands, n2, nsub, histo, ch == 1 && merged ? mad + 32 + nsub : madab, sL));
            if (!fused) {
                ShrinkArgs sa = {};
                sa.coef = Cd.bands; sa.coefL = Ld.bands; sa.sfave = sf; sa.n = n2; sa.madL = madL; sa.madab = madab;
                sa.noisevar = ccalc_dev; sa.noisevar_scale = maxNoiseVarab; sa.noisevar_ab = noisevar_ab; sa.useNoiseCCurve = useNoiseCCurve ? 1 : 0;
                HIPCHK(ctx, launch_shrink_sf(sa, nsub, true, sL));
            }
        }
        return ARTGPU_OK;
    };
    // second half: box blur of the shrink factors, coefficient update, residuals, reconstruction
    float chresidtemp = 0.f, chmaxresidtemp = 0.f;
    auto chroma_back = [&](int ch) -> int {
        DevDecomp &Cd = Cdd[ch];
        float *sf = sfc[1 + ch], *tmp = tmpc[1 + ch], *madab = mad + 32 * (1 + ch);
        int *histo = reinterpret_cast<int *>(histo_fc[1 + ch]);
        if (noisevar_abc[ch] > 0.001f && !merged) {
            if (fused) {
                // (the factors read the L coefficients as the decomposition left them: in the reference's order L comes last, with the side
                // stream the L pass has written a second band set)
                int rc2 = fused_pass(true, Cd.bands, Cd.bands, Ld.bands, madL, madab, 0, nsub, 0.f, noisevar_abc[ch]);
                if (rc2) return rc2;
            } else {
                BlurArgs bl = bl0;
                bl.src = sf; bl.dst = tmp;
                HIPCHK(ctx, launch_hblur(bl, nsub, sL));
                bl.src = tmp; bl.sfave = sf; bl.coef = Cd.bands;
                HIPCHK(ctx, launch_vblur_combine(bl, nsub, sL));
            }
        }
        if (nresi || highresi) {
            // Noise_residualAB (FTblockDN.cc:605-635, kall == 0): SQR(MadRgb) of the shrunk chroma bands, summed in level/dir order
            float host[32];
            HIPCHK(ctx, launch_mad(Cd.bands, n2, nsub, histo, madab, sL));
            HIPCHK(ctx, hipMemcpyAsync(host, madab, (size_t)nsub * sizeof(float), hipMemcpyDeviceToHost, sL));
            HIPCHK(ctx, hipStreamSynchronize(sL));
            float resid = 0.f, maxresid = 0.f;
            for (int k = 0; k < nsub; ++k) {
                resid += host[k];
                if (host[k] > maxresid) maxresid = host[k];
            }
            if (ch == 0) { chresidtemp = resid; chmaxresidtemp = maxresid; }
            else {
                float chresid = resid + chresidtemp, chmaxresid = maxresid + chmaxresidtemp;      // L2389-2396
                chresid = std::sqrt(chresid / (6 * (levwav)));
                if (highresi) *highresi = chresid + 0.66f * (std::sqrt(chmaxresid) - chresid);
                if (nresi) *nresi = chresid;
            }
        }
        return reconstruct_dev(ctx, Cd, ch == 0 ? A : B, sL);
    };

    // ---- L: shrink the first min(levels,5) levels, reconstruct (L2405-2438); detail recovery on stream `sd`
    float *Lout = L;
    auto luma = [&](hipStream_t sd) -> int {
        if (!denoiseLuminance) return ARTGPU_OK;
        int rc2;
        const int nsubL = 3 * (levwav < 5 ? levwav : 5);
        float *sf = sfc[0], *tmp = tmpc[0];
        BlurArgs bl = bl0;
        ShrinkArgs sa = {};
        sa.coef = Ld.bands; sa.sfave = sf; sa.n = n2; sa.madL = madL; sa.noisevar = nullptr; sa.noisevar_const = noisevarL;
        // QUALITY_HIGH runs WaveletDenoiseAll_BiShrinkL first (L842-973); its per-band body is ShrinkAllL's (top level included),
        // and madL is not recomputed in between (L2408-2421): the standard pass simply runs twice
        DevDecomp Lrec = Ld;                   // what the reconstruction reads
        if (merged) Lrec.bands = Lbands2;
        for (int rep = aggressive ? 0 : 1; rep < 2 && !merged; ++rep) {
            if (fused) {
                // the first pass reads the decomposition; with a second band set it writes there, and a second pass (QUALITY_HIGH) works on that
                float *dstb = Lbands2 ? Lbands2 : Ld.bands;
                if ((rc2 = fused_pass(false, Lrec.bands, dstb, nullptr, madL, nullptr, 0, nsubL, noisevarL, 0.f))) return rc2;
                Lrec.bands = dstb;
            } else {
                HIPCHK(ctx, launch_shrink_sf(sa, nsubL, false, sL));
                bl.src = sf; bl.dst = tmp;
                HIPCHK(ctx, launch_hblur(bl, nsubL, sL));
                bl.src = tmp; bl.sfave = sf; bl.coef = Ld.bands;
                HIPCHK(ctx, launch_vblur_combine(bl, nsubL, sL));
            }
        }
        if (Lrec.bands != Ld.bands && nsubL < nsub)          // levels beyond the fifth are reconstructed as they are
            HIPCHK(ctx, hipMemcpyAsync(Lrec.bands + (size_t)nsubL * n2, Ld.bands + (size_t)nsubL * n2, (size_t)(nsub - nsubL) * n2 * 4, hipMemcpyDeviceToDevice, sL));
        if (do_detail) {
            // labdn->L is kept as Lin before the reconstruction modifies it (L2423-2432): here the reconstruction writes a second plane
            // instead of the first being copied
            if ((rc2 = pool_get(ctx, P_LIN, n * 4, &Lout))) return rc2;
        }
        Lrec.cur = Ld.cur;
        if ((rc2 = reconstruct_dev(ctx, Lrec, Lout, sL))) return rc2;
        if (do_detail) {
            float *Lin = L;
            // ---- detail_recovery (L1479-1635): host-side tables exactly as the reference builds them
            DetailArgs da = {};
            da.w = w; da.h = h;
            da.numblox_W = (int)std::ceil(((float)w) / 25) + 2;
            da.numblox_H = (int)std::ceil(((float)h) / 25) + 2;
            const float params_Ldetail = std::min(float(p->luminance_detail), 99.9f);
            auto compute_detail = [](float dd) -> float { const float t = static_cast<float>((100. - dd) * (100. - dd) + 50. * (100. - dd)) * 64 * 0.5f; return t * t; };
            da.detail_hi = compute_detail(params_Ldetail);
            da.detail_lo = compute_detail(0.f);
            { const int br = int(3 / scale); da.blur_rad = br > 1 ? br : 1; }
            float *dtab;
            // the tables are constants: built and uploaded once per context (a slot of their own: the upload needs a stream
            // synchronisation, i.e. a bubble in the middle of every frame)
            const bool fresh_dtab = ctx->pool[P_DCTTAB] == nullptr;
            if ((rc2 = pool_get(ctx, P_DCTTAB, 4 * 4096 * 4, &dtab))) return rc2;
            if ((rc2 = pool_get(ctx, P_BLOCKS, (size_t)da.numblox_W * da.numblox_H * 4096 * 4, &da.blocks))) return rc2;
            if (fresh_dtab) {
                std::vector<float> host(4 * 4096);
                float *tm_in = host.data(), *tm_out = tm_in + 4096, *ct = tm_out + 4096, *ctt = ct + 4096;
                const float epsilon = 0.001f / (64 * 64);
                const int border = 4; // MAX(2, TS/16)
                for (int i = 0; i < 64; ++i) {
                    const float i1 = std::abs((i > 32 ? i - 64 + 1 : i));
                    const float vmask = (i1 < border ? (float)0 + (std::sin((M_PI * i1) / (2 * border)) * std::sin((M_PI * i1) / (2 * border))) : 1.0f);
                    const float vmask2 = (i1 < 2 * border ? (std::sin((M_PI * i1) / (2 * border)) * std::sin((M_PI * i1) / (2 * border))) : 1.0f);
                    for (int j = 0; j < 64; ++j) {
                        const float j1 = std::abs((j > 32 ? j - 64 + 1 : j));
                        const double sj = std::sin((M_PI * j1) / (2 * border));
                        tm_in[i * 64 + j] = (vmask * (j1 < border ? sj * sj : 1.0f)) + epsilon;
                        tm_out[i * 64 + j] = (vmask2 * (j1 < 2 * border ? sj * sj : 1.0f)) + epsilon;
                        ct[i * 64 + j] = (float)std::cos(M_PI * (j + 0.5) * i / 64.0);
                        ctt[j * 64 + i] = ct[i * 64 + j];
                    }
                }
                hipError_t e = hipMemcpyAsync(dtab, host.data(), host.size() * 4, hipMemcpyHostToDevice, sL);
                if (e == hipSuccess) e = hipStreamSynchronize(sL); // host vector goes out of scope
                if (e != hipSuccess) {     // (see P_LABTABS above)
                    (void)hipFree(ctx->pool[P_DCTTAB]); ctx->pool[P_DCTTAB] = nullptr; ctx->pool_bytes[P_DCTTAB] = 0;
                    return fail(ctx, ARTGPU_EHIP, "rgb_denoise: upload of the DCT tables failed: %s", hipGetErrorString(e));
                }
            }
            da.tm_in = dtab; da.tm_out = dtab + 4096; da.costab = dtab + 2 * 4096; da.costab_t = dtab + 3 * 4096;
            da.L = Lout; da.Lin = Lin;
            if (p->luminance_detail_threshold > 0) {
                // detail_mask(LL, mask, 65535, 25, 10000, amount, GAUSS, 25 / scale) on the denoised L (FTblockDN.cc:1502-1507)
                float *dmask;
                if ((rc2 = pool_get(ctx, P_DMASK, n * 4, &dmask))) return rc2;
                const float amount = std::max(0.f, std::min(float(p->luminance_detail_threshold) / 100.f, 1.f));
                float *dm_scratch = tmp;
                if (!dm_scratch && (rc2 = pool_get(ctx, P_TMP, ((size_t)w * h + 2 * (size_t)(w / 4) * (h / 4)) * 4, &dm_scratch))) return rc2;   // (fused: no `tmp` plane set)
                if ((rc2 = detail_mask_dev(ctx, Lout, (size_t)w, dmask, w, h, 65535.f, 25.f, 10000.f, amount, (float)(25.f / scale), dm_scratch))) return rc2;
                da.mask = dmask; da.params_Ldetail = params_Ldetail;
            }
            if (sd != sL) {
                HIPCHK(ctx, hipEventRecord(ctx->dn_ev[0], sL));
                HIPCHK(ctx, hipStreamWaitEvent(sd, ctx->dn_ev[0], 0));
            }
            HIPCHK(ctx, launch_detail_blocks(da, sd));
            HIPCHK(ctx, launch_detail_gather(da, sd));
            if (sd != sL) HIPCHK(ctx, hipEventRecord(ctx->dn_ev[1], sd));
        }
        return ARTGPU_OK;
    };

    auto merged_pass = [&]() -> int {
        FusedShrinkArgs fa = {};
        fa.coef = Ld.bands; fa.coef_out = Lbands2; fa.coefC = Cdd[0].bands; fa.coefL = Ld.bands; fa.n = n2; fa.w = w2; fa.h = h2;
        fa.madL = madL; fa.madab = merged_mad ? mad + nsub : mad + 32; fa.mad_ch_stride = nsub;
        fa.noisevar = ccalc_dev; fa.noisevar_nonneg = ctx->ccalc_nonneg; fa.noisevar_const = noisevarL; fa.noisevar_scale = maxNoiseVarab;
        fa.noisevar_ab[0] = noisevar_abc[0]; fa.noisevar_ab[1] = noisevar_abc[1];
        fa.useNoiseCCurve = useNoiseCCurve ? 1 : 0;
        for (int l = 0; l < 10; ++l) fa.rad[l] = bl0.rad[l];
        fa.level0 = 0; fa.nsub = 3 * nsub; fa.nL = nsub; fa.nsub_ch = nsub; fa.diag = ctx->fs_diag; fa.wait_ticks = (long long)ctx->opt_dn_wait_ms * 100000LL; fa.stall_band = ctx->opt_dn_debug_stall < 0 ? -1 : ctx->opt_dn_debug_stall >> 16; fa.stall_strip = ctx->opt_dn_debug_stall < 0 ? -1 : ctx->opt_dn_debug_stall & 0xffff;
        HIPCHK(ctx, launch_shrink_blur(fa, fused_scratch, sL));
        return ARTGPU_OK;
    };
    if (merged) {
        // decompositions and MADs of a and b, the three channels' ShrinkAll passes as one launch, then the reconstructions -- L first, so that
        // its DCT detail recovery (side stream) runs beside those of a and b
        if ((rc = chroma_front(0)) || (rc = chroma_front(1))) return rc;
        // MadRgb of all three channels' bands as one launch set (the bands are back to back: L, a, b; the medians land at mad + band)
        if (merged_mad) HIPCHK(ctx, launch_mad(Ld.bands, n2, 3 * nsub, reinterpret_cast<int *>(histo_fc[0]), mad, sL));
        if ((rc = merged_pass())) return rc;
        SideStreamJoin dn_join;
        if (fork) dn_join.arm(ctx->dn_stream[0]);
        if ((rc = luma(fork ? ctx->dn_stream[0] : sL))) return rc;
        if ((rc = chroma_back(0)) || (rc = chroma_back(1))) return rc;
        if (fork) {
            HIPCHK(ctx, hipStreamWaitEvent(sL, ctx->dn_ev[1], 0));
            dn_join.disarm();
        }
    } else if (!fork) {
        // the reference's order
        for (int ch = 0; ch < 2; ++ch)
            if ((rc = chroma_front(ch)) || (rc = chroma_back(ch))) return rc;
        if ((rc = luma(sL))) return rc;
    } else {
        if ((rc = chroma_front(0)) || (rc = chroma_front(1))) return rc;      // the last readers of the untouched L coefficients
        SideStreamJoin dn_join;
        dn_join.arm(ctx->dn_stream[0]);                                       // any return below leaves with the side stream drained
        if ((rc = luma(ctx->dn_stream[0]))) return rc;
        if ((rc = chroma_back(0)) || (rc = chroma_back(1))) return rc;
        HIPCHK(ctx, hipStreamWaitEvent(sL, ctx->dn_ev[1], 0));                // join: the context's stream is behind all the work of the call
        dn_join.disarm();
    }
    px.L = Lout;

    // ---- back to RGB (L2502-2550)
    HIPCHK(ctx, launch_yuv2rgb(px, ctx->stream));
    return unbind_rgb(ctx, img, &d);
}

// ---------------------------------------------------------------------------------------------
// guided chroma smoothing
// ---------------------------------------------------------------------------------------------
namespace {
int gf_subsampling(int w, int h, int r)   // calculate_subsampling (guidedfilter.cc:58-75)
{
    if (r == 1) return 1;
    if ((w > h ? w : h) <= 600) return 1;
    for (int s = 5; s > 0; --s)
        if (r % s == 0) return s;
    const int t = r / 2;
    return t < 2 ? 2 : (t > 4 ? 4 : t);
}
}

int artgpu_denoise_guided_smoothing(artgpu_ctx *ctx, artgpu_rgb *img, const double ws[9], int guided_chroma_radius, double scale)
{
    StageScope scope_(ctx, "denoise::denoiseGuidedSmoothing");
    if (!ctx) return ARTGPU_EINVAL;
    if (!img || !ws || !(scale >= 1.0)) return fail(ctx, ARTGPU_EINVAL, "denoise_guided_smoothing: bad arguments");
    if (guided_chroma_radius == 0) return ARTGPU_OK;   // ipsmoothing.cc:877-879
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevRGB d;
    int rc = bind_rgb(ctx, img, 4, true, &d, "denoise_guided_smoothing");
    if (rc) return rc;
    const int W = d.w, H = d.h;
    int r = (int)std::round(guided_chroma_radius / scale);
    r = r > 0 ? r : 0;
    GuidedArgs g = {};
    for (int k = 0; k < 3; ++k) { g.rgb[k] = d.p[k]; g.ws1[k] = ws[3 + k]; }
    g.stride = d.stride; g.W = W; g.H = H; g.epsilon = 0.001f;
    const int sub = gf_subsampling(W, H, r);
    g.w = W / sub; g.h = H / sub;
    if (r == 0 || g.w < 8 || g.h < 8) return fail(ctx, ARTGPU_EUNSUPPORTED, "denoise_guided_smoothing: radius/scale combination not on the device path");
    const size_t n = (size_t)W * H, nl = (size_t)g.w * g.h;
    float *big, *low, *tmp;
    if ((rc = pool_get(ctx, P_SF, 7 * n * 4, &big)) || (rc = pool_get(ctx, P_TMP, 8 * nl * 4, &low)) || (rc = pool_get(ctx, P_LIN, 8 * nl * 4, &tmp))) return rc;
    for (int k = 0; k < 3; ++k) { g.in[k] = big + k * n; g.chan[k] = big + (3 + k) * n; }
    g.guide = big + 6 * n;
    for (int k = 0; k < 8; ++k) g.low[k] = low + k * nl;
    HIPCHK(ctx, launch_gf_prepare(g, ctx->stream));
    HIPCHK(ctx, launch_gf_subsample(g, ctx->stream));
    // f_mean (guidedfilter.cc:160-167): rad = LIM(int(r1), 0, (min(w,h)-1)/2 - 1); boxblur.h:318 variant
    const float r1 = float(r) / sub;
    int rad = (int)r1;
    { const int hi = ((g.w < g.h ? g.w : g.h) - 1) / 2 - 1; rad = rad < hi ? rad : hi; rad = rad > 0 ? rad : 0; }
    BlurArgs bl = {};
    bl.n = nl; bl.w = g.w; bl.h = g.h; bl.steady_div = 1; bl.plain = 1;
    for (int l = 0; l < 10; ++l) bl.rad[l] = rad;
    auto blur_planes = [&](float *planes, int count) -> int {
        if (rad == 0) return ARTGPU_OK;
        bl.src = planes; bl.dst = tmp;
        HIPCHK(ctx, launch_hblur(bl, count, ctx->stream));
        bl.src = tmp; bl.dst = planes; bl.sfave = nullptr; bl.coef = nullptr;
        HIPCHK(ctx, launch_vblur_combine(bl, count, ctx->stream));
        return ARTGPU_OK;
    };
    if ((rc = blur_planes(low, 8))) return rc;                 // meanI, corrI, meanp[3], corrIp[3]
    HIPCHK(ctx, launch_gf_ab(g, ctx->stream));
    if ((rc = blur_planes(low + 2 * nl, 6))) return rc;        // mean a[3], mean b[3]
    HIPCHK(ctx, launch_gf_finish(g, ctx->stream));
    return unbind_rgb(ctx, img, &d);
}

int artgpu_guided_filter(artgpu_ctx *ctx, const artgpu_plane *guide, const artgpu_plane *src, artgpu_plane *dst, int r, float epsilon, int subsampling)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!guide || !src || !dst || !plane_ok(guide) || !plane_ok(src) || !plane_ok(dst)) return fail(ctx, ARTGPU_EINVAL, "guided_filter: bad plane");
    const int W = src->w, H = src->h;
    if (guide->w != W || guide->h != H || dst->w != W || dst->h != H || r < 0) return fail(ctx, ARTGPU_EINVAL, "guided_filter: size mismatch / negative radius");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int sub = subsampling > 0 ? subsampling : gf_subsampling(W, H, r);
    GuidedArgs g = {};
    g.W = W; g.H = H; g.w = W / sub; g.h = H / sub; g.epsilon = epsilon; g.nch = 1;
    if (g.w < 1 || g.h < 1) return fail(ctx, ARTGPU_EINVAL, "guided_filter: subsampling %d leaves no pixels", sub);
    const size_t n = (size_t)W * H, nl = (size_t)g.w * g.h;
    float *big, *low, *tmp;
    int rc;
    if ((rc = pool_get(ctx, P_SF, 3 * n * 4, &big)) || (rc = pool_get(ctx, P_TMP, 8 * nl * 4, &low)) || (rc = pool_get(ctx, P_LIN, 8 * nl * 4, &tmp))) return rc;
    g.guide = big; g.chan[0] = big + n; g.q = big + 2 * n; g.q_stride = W;
    const size_t rowb = (size_t)W * 4;
    HIPCHK(ctx, hipMemcpy2DAsync(g.guide, rowb, guide->p, (size_t)guide->row_stride_bytes, rowb, H, guide->on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpy2DAsync(g.chan[0], rowb, src->p, (size_t)src->row_stride_bytes, rowb, H, src->on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    for (int k = 0; k < 8; ++k) g.low[k] = low + k * nl;
    HIPCHK(ctx, launch_gf_subsample(g, ctx->stream));
    const float r1 = float(r) / sub;
    int rad = (int)r1;
    { const int hi = ((g.w < g.h ? g.w : g.h) - 1) / 2 - 1; rad = rad < hi ? rad : hi; rad = rad > 0 ? rad : 0; }     // f_mean's LIM (L160-164)
    if (rad > HBLUR_MAX_RADIUS) return fail(ctx, ARTGPU_EUNSUPPORTED, "guided_filter: box radius %d (r / subsampling) is above the %d the blur kernels hold in LDS", rad, HBLUR_MAX_RADIUS);
    BlurArgs bl = {};
    bl.n = nl; bl.w = g.w; bl.h = g.h; bl.steady_div = 1; bl.plain = 1;
    for (int l = 0; l < 10; ++l) bl.rad[l] = rad;
    auto blur_plane = [&](float *plane) -> int {
        if (rad == 0) return ARTGPU_OK;
        bl.src = plane; bl.dst = tmp;
        HIPCHK(ctx, launch_hblur(bl, 1, ctx->stream));
        bl.src = tmp; bl.dst = plane; bl.sfave = nullptr; bl.coef = nullptr;
        HIPCHK(ctx, launch_vblur_combine(bl, 1, ctx->stream));
        return ARTGPU_OK;
    };
    // low[0] I1 -> meanI, low[1] I1*I1 -> corrI, low[2] p1 -> meanp, low[5] I1*p1 -> corrIp
    if ((rc = blur_plane(g.low[0])) || (rc = blur_plane(g.low[1])) || (rc = blur_plane(g.low[2])) || (rc = blur_plane(g.low[5]))) return rc;
    HIPCHK(ctx, launch_gf_ab(g, ctx->stream));
    if ((rc = blur_plane(g.low[2])) || (rc = blur_plane(g.low[5]))) return rc;       // mean a, mean b
    HIPCHK(ctx, launch_gf_finish_plain(g, ctx->stream));
    HIPCHK(ctx, hipMemcpy2DAsync(dst->p, (size_t)dst->row_stride_bytes, g.q, rowb, rowb, H, dst->on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    if (!dst->on_device) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ARTGPU_OK;
}

static int lab_tabs_dev(artgpu_ctx *ctx, float **tabs_out);
// RawImageSource::vng4_demosaic on device planes (vng4_demosaic_RT.cc:62-397)
static int vng4_dev(artgpu_ctx *ctx, const float *raw, size_t raw_stride, float *r, float *g, float *b, size_t out_stride, int W, int H, uint32_t filters)
{
    if (W < 16 || H < 16) return fail(ctx, ARTGPU_EUNSUPPORTED, "vng4: image %dx%d smaller than 16x16", W, H);
    float *image, *codef;
    int rc;
    if ((rc = pool_get(ctx, P_BLOCKS, (size_t)W * H * 16, &image)) || (rc = pool_get(ctx, P_DTAB, 16 * VNG4_CODE_INTS * 4, &codef))) return rc;
    Vng4Args a = {};
    a.raw = raw; a.raw_stride = raw_stride; a.red = r; a.green = g; a.blue = b; a.out_stride = out_stride;
    a.image = image; a.code = reinterpret_cast<const int *>(codef); a.w = W; a.h = H; a.filters = filters; a.prefilters = vng4_prefilters(filters);
    std::vector<int> codes(16 * VNG4_CODE_INTS, 0);
    vng4_build_code(a.prefilters, W, codes.data());
    HIPCHK(ctx, hipMemcpyAsync(codef, codes.data(), codes.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));     // `codes` goes out of scope
    HIPCHK(ctx, launch_vng4(a, ctx->stream));
    return ARTGPU_OK;
}
// dual_demosaic_RT.cc:99-152 after the first demosaicer: L, buildBlendMask (rt_algo.cc:315-498), bilinear blend
static int dual_blend_dev(artgpu_ctx *ctx, const DevImage &d, int W, int H, uint32_t filters, DualReq *req)
{
    const size_t n = (size_t)W * H;
    float *tabs, *L, *blend, *var;
    int rc;
    if ((rc = lab_tabs_dev(ctx, &tabs)) || (rc = pool_get(ctx, P_DMASK, n * 4, &L)) || (rc = pool_get(ctx, P_CCMAP, n * 4, &blend))) return rc;
    DualArgs a = {};
    a.rgb[0] = d.r; a.rgb[1] = d.g; a.rgb[2] = d.b; a.stride = d.out_stride;
    a.raw = d.raw; a.raw_stride = d.raw_stride; a.w = W; a.h = H; a.filters = filters;
    a.cachefy = tabs + 65536; a.L = L; a.blend = blend;
    HIPCHK(ctx, launch_rgb2l(a, ctx->stream));
    float thr = (float)(*req->contrast / 100.0);
    if (req->auto_contrast) {
        // the reference's two-pass search for the flattest tile (rt_algo.cc:317-432): tile statistics on the device, the serial
        // first-minimum scan over them on the host
        std::vector<float> host;
        float *res;
        auto stats = [&](int nH, int nW, int y0, int x0, int step, int ts, int *mi, int *mj, float *minvar) -> int {
            *mi = *mj = 0; *minvar = INFINITY;
            if (nH <= 0 || nW <= 0) return ARTGPU_OK;
            const size_t cnt = (size_t)nH * nW;
            int rc2 = pool_get(ctx, P_TMP, cnt * 4, &var);
            if (rc2) return rc2;
            HIPCHK(ctx, launch_tile_stats(a, nH, nW, y0, x0, step, ts, var, ctx->stream));
            host.resize(cnt);
            HIPCHK(ctx, hipMemcpyAsync(host.data(), var, cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            for (int i = 0; i < nH; ++i)
                for (int j = 0; j < nW; ++j)
                    if (host[(size_t)i * nW + j] < *minvar) { *minvar = host[(size_t)i * nW + j]; *mi = i; *mj = j; }
            return ARTGPU_OK;
        };
        auto threshold_of = [&](int ty, int tx, int ts, float *out_thr) -> int {
            int rc2 = pool_get(ctx, P_MAD, 64, &res);
            if (rc2) return rc2;
            HIPCHK(ctx, launch_contrast_threshold(a, ty, tx, ts, res, ctx->stream));
            HIPCHK(ctx, hipMemcpyAsync(out_thr, res, 4, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            return ARTGPU_OK;
        };
        for (int pass = 0; pass < 2; ++pass) {
            const int ts = 80 / (pass + 1), skip = pass == 0 ? ts : ts / 4;
            const int nW = W / skip - 3 * pass, nH = H / skip - 3 * pass;
            int mi, mj; float minvar;
            if ((rc = stats(nH, nW, 0, 0, skip, ts, &mi, &mj, &minvar))) return rc;
            if (minvar <= 1.f || pass == 1) {
                const int minY = skip * mi, minX = skip * mj;
                if (pass == 0) {
                    if ((rc = threshold_of(minY, minX, ts, &thr))) return rc;
                    break;
                }
                const int y0 = std::max(minY - skip, 0), x0 = std::max(minX - skip, 0);
                const int y1 = std::min(minY + skip, H - ts), x1 = std::min(minX + skip, W - ts);
                int mi2, mj2; float minvar2;
                if ((rc = stats(y1 - y0 + 1, x1 - x0 + 1, y0, x0, 1, ts, &mi2, &mj2, &minvar2))) return rc;
                if (minvar2 <= 8.f) { if ((rc = threshold_of(y0 + mi2, x0 + mj2, ts, &thr))) return rc; }
                else thr = 0.f;
            }
        }
    }
    *req->contrast = thr * 100.f;
    a.threshold = thr;
    HIPCHK(ctx, launch_blend_mask(a, ctx->stream));
    if (thr != 0.f) {
        artgpu_plane bp = {blend, W, H, (int64_t)W * 4, 1};
        if ((rc = artgpu_gaussian_blur(ctx, &bp, 2.0))) return rc;      // rt_algo.cc:492
    }
    if (req->second == ARTGPU_DUAL_VNG4) {
        // vng4_demosaic into temporaries, then all three channels of every pixel (dual_demosaic_RT.cc:128-148)
        float *t;
        if ((rc = pool_get(ctx, P_SF, 3 * n * 4, &t))) return rc;
        if ((rc = vng4_dev(ctx, d.raw, d.raw_stride, t, t + n, t + 2 * n, (size_t)W, W, H, filters))) return rc;
        DevImage tmp = {d.raw, d.raw_stride, t, t + n, t + 2 * n, (size_t)W, false};
        if ((rc = launch_border(ctx, tmp, W, H, filters, 3))) return rc;
        HIPCHK(ctx, launch_dual_blend_planes(a, t, t + n, t + 2 * n, (size_t)W, ctx->stream));
        return ARTGPU_OK;
    }
    HIPCHK(ctx, launch_bilinear_blend(a, ctx->stream));
    return ARTGPU_OK;
}
static int lab_mode_switch(artgpu_ctx *ctx, artgpu_rgb *img, const double m[9], bool to_lab, const char *who)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!img || !m) return fail(ctx, ARTGPU_EINVAL, "%s: null argument", who);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    float *tabs;
    int rc = lab_tabs_dev(ctx, &tabs);
    if (rc) return rc;
    DevRGB d;
    if ((rc = bind_rgb(ctx, img, 4, true, &d, who))) return rc;
    LabArgs a = {};
    for (int k = 0; k < 3; ++k) a.img[k] = d.p[k];
    a.stride = d.stride; a.w = d.w; a.h = d.h;
    for (int k = 0; k < 9; ++k) { a.ws[k] = (float)m[k]; a.iws[k] = (float)m[k]; }
    a.cachef = tabs; a.cachefy = tabs + 65536;
    HIPCHK(ctx, to_lab ? launch_rgb_to_lab(a, ctx->stream) : launch_lab_to_rgb(a, ctx->stream));
    return unbind_rgb(ctx, img, &d);
}
int artgpu_rgb_to_lab(artgpu_ctx *ctx, artgpu_rgb *img, const double ws[9]) { return lab_mode_switch(ctx, img, ws, true, "rgb_to_lab"); }
int artgpu_lab_to_rgb(artgpu_ctx *ctx, artgpu_rgb *img, const double iws[9]) { return lab_mode_switch(ctx, img, iws, false, "lab_to_rgb"); }

int artgpu_lab_histogram(artgpu_ctx *ctx, const artgpu_rgb *img, uint32_t hist[65536])
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!img || !hist) return fail(ctx, ARTGPU_EINVAL, "lab_histogram: null argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevRGB d;
    artgpu_rgb tmp = *img;
    int rc = bind_rgb(ctx, &tmp, 4, true, &d, "lab_histogram");
    if (rc) return rc;
    float *h;
    if ((rc = pool_get(ctx, P_HISTO, 65536 * 4, &h))) return rc;
    LabArgs a = {};
    for (int k = 0; k < 3; ++k) a.img[k] = d.p[k];
    a.stride = d.stride; a.w = d.w; a.h = d.h; a.hist = reinterpret_cast<unsigned *>(h);
    HIPCHK(ctx, hipMemsetAsync(h, 0, 65536 * 4, ctx->stream));
    HIPCHK(ctx, launch_lab_hist(a, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(hist, h, 65536 * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ARTGPU_OK;
}

int artgpu_lab_adjustments(artgpu_ctx *ctx, artgpu_rgb *img, const float *lcurve, const float *acurve, const float *bcurve, float chroma)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!img || !lcurve || !acurve || !bcurve) return fail(ctx, ARTGPU_EINVAL, "lab_adjustments: null argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevRGB d;
    int rc = bind_rgb(ctx, img, 4, true, &d, "lab_adjustments");
    if (rc) return rc;
    float *luts;
    constexpr size_t NL = 32772;       // lcurve padded to a multiple of four floats
    if ((rc = pool_get(ctx, P_PIPE_R, (NL + 2 * 65536) * 4, &luts))) return rc;
    if ((rc = h2d_table(ctx, luts, lcurve, 32770 * 4)) || (rc = h2d_table(ctx, luts + NL, acurve, 65536 * 4)) || (rc = h2d_table(ctx, luts + NL + 65536, bcurve, 65536 * 4))) return rc;
    LabArgs a = {};
    for (int k = 0; k < 3; ++k) a.img[k] = d.p[k];
    a.stride = d.stride; a.w = d.w; a.h = d.h;
    a.lcurve = luts; a.acurve = luts + NL; a.bcurve = luts + NL + 65536; a.chroma = chroma;
    HIPCHK(ctx, launch_lab_adjust(a, ctx->stream));
    rc = unbind_rgb(ctx, img, &d);
    if (rc) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));     // the caller's LUTs may go out of scope
    return ARTGPU_OK;
}

int artgpu_log_encoding(artgpu_ctx *ctx, artgpu_rgb *img, const artgpu_logenc_params *p, const double ws[9], int full_width, int full_height)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!img || !p || !ws) return fail(ctx, ARTGPU_EINVAL, "log_encoding: null argument");
    if (!p->enabled) return ARTGPU_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    LogEncArgs a = {};
    a.gray = std::pow(2.f, -(float)p->gain + std::log2(0.18f));                                   // ev2gray (L119-122)
    a.shadows_range = (float)p->black_ev;
    a.dynamic_range = (float)std::max(p->white_ev - p->black_ev, 0.5);
    const float b = p->target_gray > 1 && p->target_gray < 100 && a.dynamic_range > 0
        ? logenc_find_gray((float)(std::abs(p->black_ev) / a.dynamic_range), (float)(p->target_gray / 100.f)) : 0.f;
    a.linbase = b < 0.f ? 0.f : b;
    a.satcontrol = p->satcontrol ? 1 : 0;
    {   // highlight compression (L148-156): the constants with the host's libm like the reference, the per-pixel curve on the device
        a.hlcompr = p->highlight_compression > 0 ? 1 : 0;
        const float hf = float(p->highlight_compression) / 100.f;
        a.hlcompr_factor = hf < 0.f ? 0.f : (hf > 1.f ? 1.f : hf);
        constexpr float compr_l = 1.01f, compr_t = 0.8f;
        a.compr_p = std::max(a.hlcompr_factor, 0.1f);
        a.compr_s = (compr_l - compr_t) / std::pow(std::pow((1.f - compr_t) / (compr_l - compr_t), -a.compr_p) - 1.f, 1.f / a.compr_p);
    }
    { const float bl = float(p->regularization) / 100.f; a.blend = bl < 0.f ? 0.f : (bl > 1.f ? 1.f : bl); }
    for (int k = 0; k < 3; ++k) a.ws1[k] = ws[3 + k];
    DevRGB d;
    int rc = bind_rgb(ctx, img, 4, true, &d, "log_encoding");
    if (rc) return rc;
    for (int k = 0; k < 3; ++k) a.img[k] = d.p[k];
    a.stride = d.stride; a.w = d.w; a.h = d.h;
    if (p->regularization == 0) {
        HIPCHK(ctx, launch_logenc_direct(a, ctx->stream));
    } else {
        const size_t n = (size_t)d.w * d.h;
        float *Y, *Y2;
        if ((rc = pool_get(ctx, P_DMASK, n * 4, &Y)) || (rc = pool_get(ctx, P_CCMAP, n * 4, &Y2))) return rc;
        a.Y = Y; a.Y2 = Y2;
        HIPCHK(ctx, launch_logenc_prepare(a, ctx->stream));
        const int m1 = full_width > d.w ? full_width : d.w, m2 = full_height > d.h ? full_height : d.h;
        const float radius = (m1 > m2 ? m1 : m2) / 30.f;
        artgpu_plane guide = {Y2, d.w, d.h, (int64_t)d.w * 4, 1}, ypl = {Y, d.w, d.h, (int64_t)d.w * 4, 1};
        if ((rc = artgpu_guided_filter(ctx, &guide, &ypl, &ypl, (int)radius, 0.005f, 0))) return rc;
        HIPCHK(ctx, launch_logenc_blend(a, ctx->stream));
    }
    return unbind_rgb(ctx, img, &d);
}

int artgpu_hsl_equalizer(artgpu_ctx *ctx, artgpu_rgb *img, const double *hcurve, int nh, const double *scurve, int ns,
                         const double *lcurve, int nl, int smoothing, const double ws[9], double scale, int to_rgb)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!img || !ws || !(scale >= 1.0)) return fail(ctx, ARTGPU_EINVAL, "hsl_equalizer: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // the four polylines: saturation, luminance, hue (periodic, 1000 / scale points) and the fixed coefficient curve of L125-129
    static const double coeff_pts[9] = {1 /* FCT_MinMaxCPoints */, 0.25, 0.0, 0.5, 0.18, 1, 1, 0, 0.35};
    const double *pts[4] = {scurve, lcurve, hcurve, coeff_pts};
    const int npts[4] = {ns, nl, nh, 9};
    std::vector<double> px[4], py[4], ps[4];
    bool active[4];
    const int ppn = (int)(1000 / scale);
    for (int k = 0; k < 4; ++k) active[k] = pts[k] && flat_curve_polyline(pts[k], npts[k], true, k == 3 ? 1000 : ppn, 0.5, px[k], py[k], ps[k]);
    if (!active[3]) return fail(ctx, ARTGPU_EHIP, "hsl_equalizer: internal curve");
    DevRGB d;
    int rc = bind_rgb(ctx, img, 4, true, &d, "hsl_equalizer");
    if (rc) return rc;
    size_t total = 0;
    for (int k = 0; k < 4; ++k) if (active[k]) total += px[k].size() * 2 + ps[k].size();
    float *tabf, *mask;
    if ((rc = pool_get(ctx, P_PIPE_G, total * 8 + 64, &tabf)) || (rc = pool_get(ctx, P_DMASK, (size_t)d.w * d.h * 4, &mask))) return rc;
    std::vector<double> host(total);
    HslArgs a = {};
    {
        size_t off = 0;
        double *dev = reinterpret_cast<double *>(tabf);
        for (int k = 0; k < 4; ++k) {
            if (!active[k]) continue;
            const size_t n = px[k].size();
            std::copy(px[k].begin(), px[k].end(), host.begin() + off); a.curve[k].x = dev + off; off += n;
            std::copy(py[k].begin(), py[k].end(), host.begin() + off); a.curve[k].y = dev + off; off += n;
            std::copy(ps[k].begin(), ps[k].end(), host.begin() + off); a.curve[k].slope = dev + off; off += ps[k].size();
            a.curve[k].n = (int)n;
        }
        HIPCHK(ctx, hipMemcpyAsync(dev, host.data(), total * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    for (int k = 0; k < 3; ++k) { a.img[k] = d.p[k]; a.ws1[k] = (float)ws[3 + k]; }
    a.stride = d.stride; a.w = d.w; a.h = d.h; a.mask = mask; a.to_rgb = to_rgb ? 1 : 0;
    HIPCHK(ctx, launch_hsl_prepare(a, ctx->stream));
    const float sm = smoothing / 10.f;
    const float smooth = std::pow(10.f, sm < 0.f ? 0.f : (sm > 1.f ? 1.f : sm)) - 1.f;            // L93
    const int radius_small = (int)(4 / scale * smooth + 0.5), radius_large = (int)(25 / scale * smooth + 0.5);
    artgpu_plane guide = {d.p[1], d.w, d.h, (int64_t)d.stride * 4, 1};                             // Y
    artgpu_plane mpl = {mask, d.w, d.h, (int64_t)d.w * 4, 1};
    const int order[3] = {0, 1, 2};                 // saturation, luminance, hue: the reference's order
    for (int k : order) {
        if (!active[k]) continue;
        a.which = k;
        HIPCHK(ctx, launch_hsl_mask(a, ctx->stream));
        const int radius = k == 1 ? radius_large : radius_small;
        const float eps = k == 1 ? 0.0001f : 0.001f;
        if (radius > 0 && (rc = artgpu_guided_filter(ctx, &guide, &mpl, &mpl, radius, eps, 0))) return rc;
        HIPCHK(ctx, launch_hsl_apply(a, ctx->stream));
    }
    HIPCHK(ctx, launch_hsl_finish(a, ctx->stream));
    rc = unbind_rgb(ctx, img, &d);
    if (rc) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // host vector with the polylines goes out of scope
    return ARTGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// gaussian blur, detail mask, NL-means
// ---------------------------------------------------------------------------------------------
namespace {

// calculateYvVFactors<double> + M rescaling (gauss.cc:94-126,556-562)
void yvv_factors(double sigma, GaussArgs &g, bool double_path = false)
{
    double q;
    if (sigma < 2.5) q = 3.97156 - 4.14554 * std::sqrt(1.0 - 0.26891 * sigma);
    else q = 0.98711 * sigma - 0.96330;
    const double b0 = 1.57825 + 2.44413 * q + 1.4281 * q * q + 0.422205 * q * q * q;
    double b1 = 2.44413 * q + 2.85619 * q * q + 1.26661 * q * q * q;
    double b2 = -1.4281 * q * q - 1.26661 * q * q * q;
    double b3 = 0.422205 * q * q * q;
    g.B = 1.0 - (b1 + b2 + b3) / b0;
    b1 /= b0; b2 /= b0; b3 /= b0;
    double *M = g.M;
    M[0] = -b3 * b1 + 1.0 - b3 * b3 - b2;
    M[1] = (b3 + b1) * (b2 + b3 * b1);
    M[2] = b3 * (b1 + b3 * b2);
    M[3] = b1 + b3 * b2;
    M[4] = -(b2 - 1.0) * (b2 + b3 * b1);
    M[5] = -(b3 * b1 + b3 * b3 + b2 - 1.0) * b3;
    M[6] = b3 * b1 + b2 + b1 * b1 - b2 * b2;
    M[7] = b1 * b2 + b3 * b2 * b2 - b1 * b3 * b3 - b3 * b3 * b3 - b3 * b2 + b3;
    M[8] = b3 * (b1 + b3 * b2);
    for (int i = 0; i < 9; ++i) {
        if (double_path) {           // gaussHorizontal<T> / gaussVertical<T> normalise differently (gauss.cc:674-677 vs 559-563)
            M[i] /= (1.0 + b1 - b2 + b3) * (1.0 + b2 + (b1 - b3) * b3);
        } else {
            M[i] *= (1.0 + b2 + (b1 - b3) * b3);
            M[i] /= (1.0 + b1 - b2 + b3) * (1.0 - b1 - b2 - b3);
        }
        g.Mf[i] = (float)M[i];
    }
    g.b[0] = b1; g.b[1] = b2; g.b[2] = b3;
    g.Bf = (float)g.B; g.bf[0] = (float)b1; g.bf[1] = (float)b2; g.bf[2] = (float)b3;
}

// contiguous device working copy of one plane (host planes are staged, strided device planes copied)
int plane_to_pool(artgpu_ctx *ctx, const artgpu_plane *pl, int slot, float **out)
{
    const size_t rowb = (size_t)pl->w * 4;
    int rc = pool_get(ctx, slot, rowb * pl->h, out);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpy2DAsync(*out, rowb, pl->p, (size_t)pl->row_stride_bytes, rowb, pl->h,
                                 pl->on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    return ARTGPU_OK;
}
int pool_to_plane(artgpu_ctx *ctx, const float *src, artgpu_plane *pl)
{
    const size_t rowb = (size_t)pl->w * 4;
    HIPCHK(ctx, hipMemcpy2DAsync(pl->p, (size_t)pl->row_stride_bytes, src, rowb, rowb, pl->h,
                                 pl->on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    if (!pl->on_device) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ARTGPU_OK;
}

int gaussian_dev(artgpu_ctx *ctx, float *img, float *tmp, int W, int H, double sigma)
{
    // gaussianBlur's dispatch for src == dst (gauss.cc:1436-1523; the box-blur approximation above it is only taken for
    // sigma > 2 of the frame width, far off this path)
    if (!(sigma == sigma) || W < 8 || H < 8) return fail(ctx, ARTGPU_EUNSUPPORTED, "gaussian_blur: sigma %g / size %dx%d not on the device path", sigma, W, H);
    if (sigma < 0.25) return ARTGPU_OK;                 // GAUSS_SKIP: no filtering
    if (sigma < 0.6) {                                  // GAUSS_3X3_LIMIT: separated 3-tap kernel, coefficients in double, passed as float
        double c1 = exp(-1.0 / (2.0 * sigma * sigma));
        const double csum = 2.0 * c1 + 1.0;
        c1 /= csum;
        const double c0 = 1.0 / csum;
        HIPCHK(ctx, launch_gaussian3(img, tmp, W, H, (float)c0, (float)c1, ctx->stream));
        return ARTGPU_OK;
    }
    GaussArgs g = {};
    g.img = img; g.tmp = tmp; g.W = W; g.H = H;
    if (sigma >= 25.0) {                    // GAUSS_DOUBLE (gauss.cc:1393,1520-1523): all-double recursion
        float *t64;
        int rc = pool_get(ctx, P_GAUSS64, (size_t)W * H * sizeof(double), &t64);
        if (rc) return rc;
        g.tmp64 = reinterpret_cast<double *>(t64);
        yvv_factors(sigma, g, true);
    } else {
        yvv_factors((double)(float)sigma, g);   // the Sse functions take `const float sigma`
    }
    HIPCHK(ctx, launch_gaussian(g, ctx->stream));
    return ARTGPU_OK;
}

int detail_mask_dev(artgpu_ctx *ctx, const float *src, size_t src_stride, float *mask, int W, int H,
                    float scaling, float threshold, float ceiling, float factor, float blur, float *scratch /* >= W*H + 2*(W/4)*(H/4) */)
{
    MaskArgs m = {};
    m.src = src; m.src_stride = src_stride; m.mask = mask; m.W = W; m.H = H; m.w4 = W / 4; m.h4 = H / 4;
    m.L2 = scratch + (size_t)W * H; m.m2 = m.L2 + (size_t)m.w4 * m.h4;
    m.scaling = scaling; m.threshold = threshold; m.ceiling = ceiling; m.factor = factor;
    HIPCHK(ctx, launch_detail_mask(m, ctx->stream));
    return gaussian_dev(ctx, mask, scratch, W, H, blur);
}

} // namespace

int artgpu_gaussian_blur(artgpu_ctx *ctx, artgpu_plane *img, double sigma)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!plane_ok(img)) return fail(ctx, ARTGPU_EINVAL, "gaussian_blur: bad plane");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    float *work, *tmp;
    int rc = plane_to_pool(ctx, img, P_SF, &work);
    if (rc) return rc;
    if ((rc = pool_get(ctx, P_TMP, (size_t)img->w * img->h * 4, &tmp))) return rc;
    if ((rc = gaussian_dev(ctx, work, tmp, img->w, img->h, sigma))) return rc;
    return pool_to_plane(ctx, work, img);
}

int artgpu_detail_mask(artgpu_ctx *ctx, const artgpu_plane *src, artgpu_plane *mask, float scaling, float threshold,
                       float ceiling, float factor, float blur)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!plane_ok(src) || !plane_ok(mask) || mask->w != src->w || mask->h != src->h) return fail(ctx, ARTGPU_EINVAL, "detail_mask: bad planes");
    if (src->w < 32 || src->h < 32) return fail(ctx, ARTGPU_EUNSUPPORTED, "detail_mask: image smaller than 32x32");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int W = src->w, H = src->h;
    float *in, *m, *scratch;
    int rc = plane_to_pool(ctx, src, P_SF, &in);
    if (rc) return rc;
    if ((rc = pool_get(ctx, P_TMP, (size_t)W * H * 4, &m))) return rc;
    if ((rc = pool_get(ctx, P_LIN, ((size_t)W * H + 2 * (size_t)(W / 4) * (H / 4)) * 4, &scratch))) return rc;
    if ((rc = detail_mask_dev(ctx, in, W, m, W, H, scaling, threshold, ceiling, factor, blur, scratch))) return rc;
    return pool_to_plane(ctx, m, mask);
}

int artgpu_nlmeans(artgpu_ctx *ctx, artgpu_plane *img, float normcoeff, int strength, int detail_thresh, float scale)
{
    StageScope scope_(ctx, "denoise::NLMeans");
    if (!ctx) return ARTGPU_EINVAL;
    if (!plane_ok(img) || !(scale >= 1.f)) return fail(ctx, ARTGPU_EINVAL, "nlmeans: bad arguments");
    if (!strength) return ARTGPU_OK;                       // nlmeans.cc:52-54
    if (img->w < 32 || img->h < 32) return fail(ctx, ARTGPU_EUNSUPPORTED, "nlmeans: image smaller than 32x32");
    // (the tile kernel addresses a plane through 32-bit byte offsets)
    if ((unsigned long long)(img->w + 14) * (unsigned long long)(img->h + 14) >= (1ull << 30)) return fail(ctx, ARTGPU_EUNSUPPORTED, "nlmeans: image of %dx%d exceeds 2^30 pixels", img->w, img->h);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int W = img->w, H = img->h;
    NlmArgs a = {};
    a.search_radius = int(std::ceil(5.f / scale));
    a.patch_radius = int(std::ceil(2.f / scale));
    const float t = std::pow(float(strength) / 100.f, 0.9f) / 10.f / scale;
    a.h2 = t * t;
    float amount = float(detail_thresh) / 100.f;
    amount = amount < 0.f ? 0.f : (amount > 0.99f ? 0.99f : amount);   // LIM(x, 0, 0.99)
    a.border = a.search_radius + a.patch_radius;
    a.W = W; a.H = H; a.WW = W + 2 * a.border; a.HH = H + 2 * a.border; a.factor = normcoeff;
    a.ntiles_x = int(std::ceil(float(a.WW) / (150 - 2 * a.border)));
    a.ntiles_y = int(std::ceil(float(a.HH) / (150 - 2 * a.border)));
    float *work, *scratch, *pad;
    // a contiguous device plane is worked on where it lies (the kernels read the padded copy `pad` and the mask, and write the plane): two plane
    // copies less per call, 0.16 ms of a 45 MP frame; anything else goes through a contiguous working copy
    const bool in_place = img->on_device && img->row_stride_bytes == (int64_t)W * 4;
    int rc = ARTGPU_OK;
    if (in_place) work = img->p;
    else if ((rc = plane_to_pool(ctx, img, P_SF, &work))) return rc;
    if ((rc = pool_get(ctx, P_TMP, (size_t)W * H * 4 * 2 + 8192 * 4, &a.mask))) return rc;
    a.SW = a.mask + (size_t)W * H; a.explut = a.SW + (size_t)W * H;
    if ((rc = pool_get(ctx, P_LIN, ((size_t)W * H + 2 * (size_t)(W / 4) * (H / 4)) * 4, &scratch))) return rc;
    if ((rc = pool_get(ctx, P_BLOCKS, (size_t)a.WW * a.HH * 4, &pad))) return rc;
    if ((rc = detail_mask_dev(ctx, work, W, a.mask, W, H, normcoeff, 1e-3f * normcoeff, normcoeff, amount, 2.f / scale, scratch))) return rc;
    a.img = work; a.img_stride = W; a.src = pad;
    HIPCHK(ctx, launch_nlm(a, ctx->stream));
    return in_place ? ARTGPU_OK : pool_to_plane(ctx, work, img);
}

// ---------------------------------------------------------------------------------------------
// X-Trans demosaic
// ---------------------------------------------------------------------------------------------
int artgpu_demosaic_xtrans(artgpu_ctx *ctx, int passes, int use_cielab, const artgpu_plane *raw, const int32_t xtrans[36],
                           const float rgb_cam[12], artgpu_rgb *out)
{
    StageScope scope_(ctx, "RawImageSource::xtrans_interpolate");
    if (!ctx) return ARTGPU_EINVAL;
    if (!plane_ok(raw) || !out || !xtrans || !rgb_cam) return fail(ctx, ARTGPU_EINVAL, "demosaic_xtrans: bad raw plane or null argument");
    if (passes < 1 || passes > 4) return fail(ctx, ARTGPU_EINVAL, "demosaic_xtrans: passes must be 1..4");
    const int W = raw->w, H = raw->h;
    if (W < 64 || H < 64) return fail(ctx, ARTGPU_EUNSUPPORTED, "demosaic_xtrans: image %dx%d smaller than 64x64", W, H);
    int ngreen = 0;
    for (int i = 0; i < 36; ++i) {
        if (xtrans[i] < 0 || xtrans[i] > 2) return fail(ctx, ARTGPU_EINVAL, "demosaic_xtrans: colour map entries must be 0..2");
        ngreen += xtrans[i] == 1;
    }
    if (ngreen != 20) return fail(ctx, ARTGPU_EUNSUPPORTED, "demosaic_xtrans: not an X-Trans colour map (%d green of 36)", ngreen);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevImage d;
    int rc = bind_images(ctx, raw, out, &d);
    if (rc) return rc;
    // the reference keeps the hexagon offsets in `short` (xtrans_demosaic.cc:233): same range limit here
    if (2 * (long long)d.raw_stride + 2 > 32767) return fail(ctx, ARTGPU_EUNSUPPORTED, "demosaic_xtrans: row stride %zu exceeds the reference's short offsets", d.raw_stride);

    XtransArgs a = {};
    a.raw = d.raw; a.raw_stride = d.raw_stride;
    a.red = d.r; a.green = d.g; a.blue = d.b; a.out_stride = d.out_stride;
    a.W = W; a.H = H; a.passes = passes; a.ndir = 4 << (passes > 1); a.use_cielab = use_cielab ? 1 : 0;
    for (int i = 0; i < 36; ++i) a.xtrans[i] = xtrans[i];
    auto isgreen = [&](int row, int col) { return a.xtrans[(row % 3) * 6 + col % 3] & 1; };
    {   // xyz_cam (L219-230)
        static const float xyz_rgb[3][3] = {{0.412453, 0.357580, 0.180423}, {0.212671, 0.715160, 0.072169}, {0.019334, 0.119193, 0.950227}};
        static const float d65_white[3] = {0.950456, 1, 1.088754};
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                float acc = 0;
                for (int k = 0; k < 3; k++) acc += xyz_rgb[i][k] * rgb_cam[k * 4 + j] / d65_white[i];
                a.xyz_cam[i * 3 + j] = acc;
            }
    }
    {   // green hexagons around each site class and the solitary-green phase (L232-267)
        static const int orth[12] = {1, 0, 0, 1, -1, 0, 0, -1, 1, 0, 0, 1};
        static const int patt[2][16] = {{0, 1, 0, -1, 2, 0, -1, 0, 1, 1, 1, -1, 0, 0, 0, 0}, {0, 1, 0, -2, 1, 0, -2, 0, 1, 1, -2, -2, 1, -1, -1, 1}};
        for (int row = 0; row < 3; row++)
            for (int col = 0; col < 3; col++) {
                const int gint = isgreen(row, col);
                for (int ng = 0, dd = 0; dd < 10; dd += 2) {
                    if (isgreen(row + orth[dd] + 6, col + orth[dd + 2] + 6)) ng = 0; else ng++;
                    if (ng == 4) { a.sgrow = row; a.sgcol = col; }
                    if (ng == gint + 1)
                        for (int c = 0; c < 8; c++) {
                            const int v = orth[dd] * patt[gint][c * 2] + orth[dd + 1] * patt[gint][c * 2 + 1];
                            const int h = orth[dd + 2] * patt[gint][c * 2] + orth[dd + 3] * patt[gint][c * 2 + 1];
                            a.allhex0[row][col][c ^ (gint * 2 & dd)] = h + v * (int)d.raw_stride;
                            a.allhex1[row][col][c ^ (gint * 2 & dd)] = h + v * XTRANS_TS;
                        }
                }
            }
        for (int row = 0; row < 3; row++) {
            int greencount = 0;
            for (int col = 0; col < 3; col++) greencount += isgreen(row, col);
            a.right_shift[row] = greencount == 2;
        }
    }
    float *lut;
    const bool fresh = ctx->pool[P_XCBRT] == nullptr;
    if ((rc = pool_get(ctx, P_XCBRT, 0x14000 * 4, &lut))) return rc;
    if (fresh) {   // cielab's LUT (L43-57), built on the host like the reference's
        std::vector<float> host(0x14000);
        const double eps = 216.0 / 24389.0, kappa = 24389.0 / 27.0;
        for (int i = 0; i < 0x14000; i++) {
            const double r = i / 65535.0;
            host[i] = (float)(r > eps ? std::cbrt(r) : (kappa * r + 16.0) / 116.0);
        }
        HIPCHK(ctx, hipMemcpyAsync(lut, host.data(), host.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    a.cbrt_lut = lut;
    a.ntx = (W - 22 + (XTRANS_TS - 16) - 1) / (XTRANS_TS - 16);
    const int nty = (H - 22 + (XTRANS_TS - 16) - 1) / (XTRANS_TS - 16);
    a.ntiles = a.ntx * nty;
    // one 1024-thread workgroup is resident per CU: one persistent workgroup per CU, tiles from a shared counter (the arena is 0.45 GB)
    if (ctx->num_cus <= 0) {
        hipDeviceProp_t prop;
        HIPCHK(ctx, hipGetDeviceProperties(&prop, ctx->device));
        ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int grid = a.ntiles < ctx->num_cus ? a.ntiles : ctx->num_cus;
    if (!ctx->rcd_counter) HIPCHK(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->rcd_counter), 64));
    a.counter = ctx->rcd_counter + 4;           // (the RCD kernel's counter is word 0 of the same 64 bytes)
    HIPCHK(ctx, hipMemsetAsync(a.counter, 0, sizeof(int), ctx->stream));
    a.arena_floats = (size_t)XTRANS_TS * XTRANS_TS * (a.ndir * 4 + 3) + 128;
    rc = ensure(ctx, &ctx->arena, &ctx->arena_bytes, (size_t)grid * a.arena_floats * sizeof(float));
    if (rc) return rc;
    a.arena = ctx->arena;
    if (ctx->timing) HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    HIPCHK(ctx, launch_xtrans(a, grid, ctx->stream));
    if (ctx->timing) { HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream)); HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream)); }
    rc = unbind_images(ctx, out, &d);
    if (rc) return rc;
    if (ctx->timing) {
        HIPCHK(ctx, hipEventSynchronize(ctx->ev[2]));
        HIPCHK(ctx, hipEventElapsedTime(&ctx->last.demosaic_ms, ctx->ev[0], ctx->ev[1]));
        ctx->last.border_ms = 0.f; ctx->last.total_ms = ctx->last.demosaic_ms;
    }
    return ARTGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// NEUTRAL tone curve
// ---------------------------------------------------------------------------------------------
int artgpu_tone_curve_neutral(artgpu_ctx *ctx, artgpu_rgb *image, const float *lut65536, float whitecoeff, const artgpu_neutral_state *st)
{
    StageScope scope_(ctx, "ImProcFunctions::toneCurve (NEUTRAL)");
    if (!ctx) return ARTGPU_EINVAL;
    if (!image || !lut65536 || !st || !(whitecoeff > 0.f)) return fail(ctx, ARTGPU_EINVAL, "tone_curve_neutral: null/invalid argument");
    DevRGB d;
    int rc = bind_rgb(ctx, image, 4, true, &d, "tone_curve_neutral");
    if (rc) return rc;
    float *pq;
    const bool fresh = ctx->pool[P_PQ] == nullptr;
    if ((rc = pool_get(ctx, P_PQ, (2 * 65536 + 64) * 4, &pq))) return rc;
    if (fresh) {
        std::vector<float> host(2 * 65536);
        build_pq_luts(host.data(), host.data() + 65536);
        HIPCHK(ctx, hipMemcpyAsync(pq, host.data(), host.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    if ((rc = upload_curve(ctx, lut65536))) return rc;
    NeutralArgs a = {};
    for (int k = 0; k < 3; ++k) a.img[k] = d.p[k];
    a.stride = d.stride; a.w = d.w; a.h = d.h;
    a.lut = ctx->lut; a.pq = pq; a.pq_inv = pq + 65536; a.hues = pq + 2 * 65536;
    for (int k = 0; k < 9; ++k) { a.ws[k] = (float)st->ws[k]; a.iws[k] = (float)st->iws[k]; a.to_out[k] = st->to_out[k]; a.to_work[k] = st->to_work[k]; }
    a.whitecoeff = whitecoeff;
    a.tail_kind = ctx->curve_tail_kind == ARTGPU_CURVE_TAIL_HOST ? 0 : ctx->curve_tail_kind; a.tail_y = ctx->curve_tail_y; a.tail_pc = ctx->curve_tail_pc;
    if (fresh) HIPCHK(ctx, launch_neutral_hues(a, ctx->stream));
    a.no_lds_lut = !ctx->opt_lut_lds; a.cu_reserve = ctx->cu_reserve;
    HIPCHK(ctx, launch_tone_neutral(a, ctx->stream));
    return unbind_rgb(ctx, image, &d);
}

// ---------------------------------------------------------------------------------------------
// chroma noise-curve map + ImProcFunctions::denoise
// ---------------------------------------------------------------------------------------------
int artgpu_noise_curve_lut(const double *points, int npoints, float lut[501], float *sum)
{
    if (!points || npoints < 0 || !lut) return ARTGPU_EINVAL;
    const float s = noise_curve_lut(points, npoints, lut);
    if (sum) *sum = s;
    return ARTGPU_OK;
}

// fills P_CCMAP ((w+1)/2 x (h+1)/2, contiguous) from device planes
static int chroma_map_dev(artgpu_ctx *ctx, float *const planes[3], size_t stride, int w, int h, const double *mat, const double ws[9],
                          const float *curve, float **out)
{
    const int wid = (w + 1) / 2, hei = (h + 1) / 2;
    float *map, *tab;
    int rc;
    const bool fresh = ctx->pool[P_CACHEF] == nullptr;
    if ((rc = pool_get(ctx, P_CCMAP, (size_t)wid * hei * 4, &map)) || (rc = pool_get(ctx, P_CACHEF, (65536 + 512) * 4, &tab))) return rc;
    if (fresh) {
        std::vector<float> host(65536);
        build_cachef(host.data());
        ctx->ncurve_host.clear();
        hipError_t e = hipMemcpyAsync(tab, host.data(), 65536 * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // host vector goes out of scope
        if (e != hipSuccess) {      // the slot is the "table uploaded" flag: give it back
            (void)hipFree(ctx->pool[P_CACHEF]); ctx->pool[P_CACHEF] = nullptr; ctx->pool_bytes[P_CACHEF] = 0;
            return fail(ctx, ARTGPU_EHIP, "chroma map: upload of the cachef table failed: %s", hipGetErrorString(e));
        }
    }
    // the same curve frame after frame is uploaded once: the copy needs a stream synchronisation (the caller's curve may be a
    // temporary), i.e. a bubble in the middle of every frame
    if (fresh || ctx->ncurve_host.size() != 501 || std::memcmp(ctx->ncurve_host.data(), curve, 501 * 4) != 0) {
        ctx->ncurve_host.assign(curve, curve + 501);
        hipError_t e = hipMemcpyAsync(tab + 65536, ctx->ncurve_host.data(), 501 * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            ctx->ncurve_host.clear();
            return fail(ctx, ARTGPU_EHIP, "chroma map: upload of the noise curve failed: %s", hipGetErrorString(e));
        }
    }
    ChromaMapArgs a = {};
    a.gi = ctx->fuse_gi;
    for (int k = 0; k < 3; ++k) a.src[k] = planes[k];
    a.stride = stride; a.wid = wid; a.hei = hei;
    a.has_mat = mat ? 1 : 0;
    for (int k = 0; k < 9; ++k) { a.mat[k] = mat ? mat[k] : 0.0; a.wpi[k] = (float)ws[k]; }
    a.cachef = tab; a.curve = tab + 65536; a.out = map; a.no_lds_lut = !ctx->opt_lut_lds; a.cu_reserve = ctx->cu_reserve;
    HIPCHK(ctx, launch_chroma_map(a, ctx->stream));
    *out = map;
    return ARTGPU_OK;
}

// Color::cachef / cachefy / denoiseGammaTab / denoiseIGammaTab on the device, built on the host like the reference's (color.cc:202-292)
static int lab_tabs_dev(artgpu_ctx *ctx, float **tabs_out)
{
    float *tabs;
    const bool fresh = ctx->pool[P_LABTABS] == nullptr;
    int rc = pool_get(ctx, P_LABTABS, 4 * 65536 * 4, &tabs);
    if (rc) return rc;
    if (fresh) {
        std::vector<float> host(4 * 65536);
        build_cachef(host.data()); build_cachefy(host.data() + 65536);
        build_denoise_gamma_tabs(host.data() + 2 * 65536, host.data() + 3 * 65536);
        HIPCHK(ctx, hipMemcpyAsync(tabs, host.data(), host.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    *tabs_out = tabs;
    return ARTGPU_OK;
}

int artgpu_denoise_compute_params(artgpu_ctx *ctx, const artgpu_rgb *planes, int border, const float mul[3], int do_clip,
                                  const double cam_to_work[9], const double ws[9], double chrominance_auto_factor,
                                  artgpu_denoise_info_store *store, artgpu_denoise_params *dn)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!planes || !mul || !cam_to_work || !ws || !store || !dn) return fail(ctx, ARTGPU_EINVAL, "denoise_compute_params: null argument");
    const bool automatic = dn->chrominance_method == 1;
    if (store->valid || !automatic) {                       // ipdenoise.cc:802-809
        if (automatic) {
            dn->chrominance = store->chrominance * chrominance_auto_factor;
            dn->chrominance_red_green = store->chrominance_red_green * chrominance_auto_factor;
            dn->chrominance_blue_yellow = store->chrominance_blue_yellow * chrominance_auto_factor;
        }
        return ARTGPU_OK;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevRGB src;
    int rc = bind_rgb(ctx, planes, 1, true, &src, "denoise_compute_params(planes)");
    if (rc) return rc;
    if (border < 0) return fail(ctx, ARTGPU_EINVAL, "denoise_compute_params: border %d", border);
    const int widIm = src.w - 2 * border, heiIm = src.h - 2 * border;       // getFullSize
    const int crW = widIm / 2, crH = heiIm / 2;                              // Tile_calc returns one tile: tileWskip = widIm (L836,875-876)
    if (widIm < 100 || heiIm < 100)
        return fail(ctx, ARTGPU_EUNSUPPORTED, "denoise_compute_params: %dx%d is too small for the nine-crop layout (crops start 50 px in)", widIm, heiIm);
    const int coordW[3] = {50, widIm / 2 - crW / 2, widIm - crW - 50}, coordH[3] = {50, heiIm / 2 - crH / 2, heiIm - crH - 50};
    const int wid = (crW + 1) / 2, hei = (crH + 1) / 2, levwav = 5, nsub = 3 * levwav;   // levwav = max(2, 5 - ceil(log(1)))
    const size_t n = (size_t)crW * crH, n2 = (size_t)wid * hei;
    if (!ctx->aux) {
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->aux_ev[0], hipEventDisableTiming));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->aux_ev[1], hipEventDisableTiming));
    }

    DnInfoArgs a = {};
    float *tabs, *gamlut, *histo_f, *res;
    DevDecomp Cd = {};
    Cd.w = crW; Cd.h = crH; Cd.w2 = wid; Cd.h2 = hei; Cd.n = n2; Cd.nlevels = levwav;
    if ((rc = lab_tabs_dev(ctx, &tabs)) || (rc = pool_get(ctx, P_A, n * 4, &a.A)) || (rc = pool_get(ctx, P_B, n * 4, &a.B)) ||
        (rc = pool_get(ctx, P_CBANDS, (size_t)nsub * n2 * 4, &Cd.bands)) || (rc = pool_get(ctx, P_CLOW0, n2 * 4, &Cd.low[0])) ||
        (rc = pool_get(ctx, P_CLOW1, n2 * 4, &Cd.low[1])) || (rc = pool_get(ctx, P_SF, (size_t)27 * n2 * 4, &a.maps)) ||
        (rc = pool_get(ctx, P_HISTO, (size_t)nsub * (65536 + MAD_SCRATCH_INTS_PER_BAND) * 4, &histo_f)) || (rc = pool_get(ctx, P_GAM, 2 * 65536 * 4, &gamlut)) ||
        (rc = pool_get(ctx, P_DNINFO, (9 * 32 + 9 * 8) * 4, &res)))
        return rc;
    for (int k = 0; k < 3; ++k) { a.src[k] = src.p[k]; a.mul[k] = mul[k]; }
    a.stride = src.stride; a.do_clip = do_clip ? 1 : 0;
    for (int wcr = 0; wcr < 3; ++wcr)
        for (int hcr = 0; hcr < 3; ++hcr) { a.sx[hcr * 3 + wcr] = coordW[wcr] + border; a.sy[hcr * 3 + wcr] = coordH[hcr] + border; }   // transformRect adds the border
    a.crW = crW; a.crH = crH; a.wid = wid; a.hei = hei;
    for (int k = 0; k < 9; ++k) { a.mat[k] = cam_to_work[k]; a.wp[k] = (float)ws[k]; }
    a.cachef = tabs; a.cachefy = tabs + 65536; a.gamcurve = gamlut;
    // RGB_denoise_infoGamCurve (ipdenoise.cc:209-224), raw: gamma as given
    a.gam = (float)dn->gamma; a.gamthresh = 0.001f;
    a.gamslope = (float)(std::exp(std::log(static_cast<double>(a.gamthresh)) / a.gam) / a.gamthresh);
    { const double expcomp = std::log(5.f) / std::log(2.f); a.gain = std::pow(2.0f, float(expcomp)); }     // L936, L291
    a.stats = res + 9 * 32;
    ctx->gam_tab = nullptr;                    // (the slot RGB_denoise keeps its gamma pair in)
    HIPCHK(ctx, launch_gamma_lut(gamlut, a.gam, a.gamthresh, a.gamslope, 65535.f, 32768.f, ctx->stream));

    // maps of all nine crops, then their serial statistics on the second stream while this one decomposes
    HIPCHK(ctx, launch_dninfo_maps(a, ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->aux_ev[0], ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->aux, ctx->aux_ev[0], 0));
    HIPCHK(ctx, launch_dninfo_stats(a, ctx->aux));
    HIPCHK(ctx, hipEventRecord(ctx->aux_ev[1], ctx->aux));
    int *histo = reinterpret_cast<int *>(histo_f);
    for (int k = 0; k < 9; ++k) {
        a.crop = k;
        HIPCHK(ctx, launch_dninfo_ab(a, ctx->stream));
        if ((rc = decompose_dev(ctx, Cd, a.A))) return rc;
        HIPCHK(ctx, launch_mad(Cd.bands, n2, nsub, histo, res + k * 32, ctx->stream));
        if ((rc = decompose_dev(ctx, Cd, a.B))) return rc;
        HIPCHK(ctx, launch_mad(Cd.bands, n2, nsub, histo, res + k * 32 + 16, ctx->stream));
    }
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->aux_ev[1], 0));
    float host[9 * 32 + 9 * 8];
    HIPCHK(ctx, hipMemcpyAsync(host, res, sizeof host, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));

    const bool aggressive = dn->aggressive != 0;
    float info[9][16] = {};
    for (int k = 0; k < 9; ++k) {
        float bs[6];
        dninfo_band_stats(host + k * 32, host + k * 32 + 16, nsub, aggressive, bs);
        const float *st = host + 9 * 32 + k * 8;
        int nry, nsk;
        memcpy(&nry, st + 4, 4); memcpy(&nsk, st + 5, 4);
        const int nc = (int)n2;                              // ShrinkAll_info, lvl == 1 (FTblockDN.cc:1266-1288)
        for (int j = 0; j < 5; ++j) info[k][j] = bs[j];
        info[k][5] = st[0] / nc;                             // chromina
        info[k][6] = st[1] / nc;                             // lumema
        info[k][7] = nry > 0 ? st[2] / nry : 0.f;            // redyel
        info[k][8] = nsk > 0 ? st[3] / nsk : 0.f;            // skinc
        info[k][9] = static_cast<float>(nsk) / static_cast<float>(nc);
        info[k][10] = bs[5];
    }
    float out3[3];
    dninfo_reduce(info, aggressive, store->ch_M, store->max_r, store->max_b, out3);
    store->chrominance = out3[0]; store->chrominance_red_green = out3[1]; store->chrominance_blue_yellow = out3[2];
    dn->chrominance = store->chrominance * chrominance_auto_factor;
    dn->chrominance_red_green = store->chrominance_red_green * chrominance_auto_factor;
    dn->chrominance_blue_yellow = store->chrominance_blue_yellow * chrominance_auto_factor;
    store->valid = 1;
    for (int k = 0; k < 9; ++k) memcpy(store->crop_info[k], info[k], sizeof info[k]);
    return ARTGPU_OK;
}

int artgpu_ordered_sum_f32(artgpu_ctx *ctx, const float *x, int64_t n, int on_device, float *result)
{
    if (!ctx) return ARTGPU_EINVAL;
    if ((!x && n > 0) || n < 0 || !result) return fail(ctx, ARTGPU_EINVAL, "ordered_sum_f32: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    float *res, *dev = nullptr;
    int rc = pool_get(ctx, P_DNINFO, (9 * 32 + 9 * 8) * 4, &res);
    if (rc) return rc;
    const float *src = x;
    if (!on_device && n > 0) {
        if ((rc = pool_get(ctx, P_TMP, (size_t)n * 4, &dev))) return rc;
        HIPCHK(ctx, hipMemcpyAsync(dev, x, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        src = dev;
    }
    HIPCHK(ctx, launch_ordered_sum(src, (long long)n, res, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(result, res, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ARTGPU_OK;
}

int artgpu_eval_primitive(artgpu_ctx *ctx, int prim, const void *a, const void *b, const void *c, void *out0, void *out1, int64_t n,
                          float param, const float *table, int table_size)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (prim < 0 || prim >= PRIM_COUNT || n < 0 || (n > 0 && (!a || !out0))) return fail(ctx, ARTGPU_EINVAL, "eval_primitive: bad argument");
    const bool dbl = prim == PRIM_XLOG_D || prim == PRIM_XEXP_D;
    const bool need_b = prim == PRIM_POW_F || prim == PRIM_XATAN2F || prim == PRIM_MEDIAN3 || prim == PRIM_VMINF || prim == PRIM_VMAXF || prim == PRIM_VINTPF;
    const bool need_c = prim == PRIM_MEDIAN3 || prim == PRIM_VINTPF;
    const bool lut = prim == PRIM_LUTF_SCALAR || prim == PRIM_LUTF_VECTOR;
    if ((need_b && !b) || (need_c && !c) || (prim == PRIM_XSINCOSF && !out1) || (lut && (!table || table_size < 2)))
        return fail(ctx, ARTGPU_EINVAL, "eval_primitive: primitive %d lacks an operand", prim);
    if (n == 0) return ARTGPU_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t esz = dbl ? 8 : 4, bytes = (size_t)n * esz, tbytes = lut ? (size_t)table_size * 4 : 0;
    float *buf;
    int rc = pool_get(ctx, P_TMP, 5 * bytes + tbytes + 64, &buf);
    if (rc) return rc;
    char *base = reinterpret_cast<char *>(buf);
    PrimArgs p{};
    p.prim = prim; p.n = n; p.param = param; p.table_size = table_size;
    p.a = base; p.b = base + bytes; p.c = base + 2 * bytes; p.out0 = base + 3 * bytes; p.out1 = base + 4 * bytes;
    p.table = reinterpret_cast<const float *>(base + 5 * bytes);
    HIPCHK(ctx, hipMemcpyAsync(base, a, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (need_b) HIPCHK(ctx, hipMemcpyAsync(base + bytes, b, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (need_c) HIPCHK(ctx, hipMemcpyAsync(base + 2 * bytes, c, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (lut) HIPCHK(ctx, hipMemcpyAsync(base + 5 * bytes, table, tbytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, launch_prim_eval(p, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(out0, p.out0, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (prim == PRIM_XSINCOSF) HIPCHK(ctx, hipMemcpyAsync(out1, p.out1, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ARTGPU_OK;
}

int artgpu_denoise_chroma_map(artgpu_ctx *ctx, const artgpu_rgb *img, const double *calclum_mat, const double ws[9],
                              const float noise_c_curve[501], artgpu_plane *ccalc)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!img || !ws || !noise_c_curve || !ccalc) return fail(ctx, ARTGPU_EINVAL, "denoise_chroma_map: null argument");
    DevRGB d;
    int rc = bind_rgb(ctx, img, 4, true, &d, "denoise_chroma_map");
    if (rc) return rc;
    const int wid = (d.w + 1) / 2, hei = (d.h + 1) / 2;
    if (!plane_ok(ccalc) || ccalc->w != wid || ccalc->h != hei) return fail(ctx, ARTGPU_EINVAL, "denoise_chroma_map: ccalc must be %dx%d", wid, hei);
    float *map;
    if ((rc = chroma_map_dev(ctx, d.p, d.stride, d.w, d.h, calclum_mat, ws, noise_c_curve, &map))) return rc;
    HIPCHK(ctx, hipMemcpy2DAsync(ccalc->p, (size_t)ccalc->row_stride_bytes, map, (size_t)wid * 4, (size_t)wid * 4, hei,
                                 ccalc->on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    if (!ccalc->on_device) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ARTGPU_OK;
}

int artgpu_improc_denoise(artgpu_ctx *ctx, artgpu_rgb *img, const artgpu_denoise_tool_params *p, const double ws[9], const double *iws,
                          double ecomp, double scale, const double *calclum_mat, const float *noise_c_curve, uint32_t flags)
{
    return artgpu_improc_denoise_fused(ctx, img, nullptr, p, ws, iws, ecomp, scale, calclum_mat, noise_c_curve, flags);
}

int artgpu_improc_denoise_fused(artgpu_ctx *ctx, artgpu_rgb *img, const artgpu_denoise_fusion *fu, const artgpu_denoise_tool_params *p,
                                const double ws[9], const double *iws, double ecomp, double scale, const double *calclum_mat,
                                const float *noise_c_curve, uint32_t flags)
{
    StageScope scope_(ctx, "ImProcFunctions::denoise");
    if (!ctx) return ARTGPU_EINVAL;
    if (!img || !p || !ws) return fail(ctx, ARTGPU_EINVAL, "improc_denoise: null argument");
    // what of the neighbouring stages can really live inside the tool's pixel passes: the wavelet denoise has to run (its first pass reads the
    // image, its last one writes it), on device planes; the exposure only when nothing stands between RGB_denoise and it
    bool use_curve0 = false;
    if (noise_c_curve) {
        float sum = 0.f;
        for (int i = 0; i < 501; ++i) sum += noise_c_curve[i];
        use_curve0 = sum > 5.f;                                        // (as below: NoiseCurve::getSum, FTblockDN.cc:1672)
    }
    const bool dn_runs0 = !(p->dn.luminance == 0 && p->dn.chrominance == 0 && !use_curve0);
    const bool dev_planes = img->r.on_device && (!fu || !fu->demosaiced || (fu->demosaiced->r.on_device && fu->demosaiced->g.on_device && fu->demosaiced->b.on_device));
    bool fuse_gi = fu && fu->demosaiced && dn_runs0 && dev_planes;
    const bool fuse_exp = fu && fu->exposure_enabled && dn_runs0 && dev_planes && !p->smoothing_enabled;
    // with guided smoothing / NL-means behind the wavelet denoise the tool's last pixel pass is setMode(RGB) or its own expcomp(-ecomp):
    // the exposure rides on that one
    const bool tail_exp = fu && fu->exposure_enabled && dn_runs0 && dev_planes && p->smoothing_enabled && (p->nl_strength || ecomp > 0);
    if (fu && fu->demosaiced) {
        const artgpu_rgb *dm = fu->demosaiced;
        if (!plane_ok(&dm->r) || !plane_ok(&dm->g) || !plane_ok(&dm->b) || dm->g.row_stride_bytes != dm->r.row_stride_bytes || dm->b.row_stride_bytes != dm->r.row_stride_bytes ||
            fu->sx1 < 0 || fu->sy1 < 0 || fu->sx1 + img->r.w > dm->r.w || fu->sy1 + img->r.h > dm->r.h)
            return fail(ctx, ARTGPU_EINVAL, "improc_denoise_fused: crop %dx%d+%d+%d outside the demosaiced planes", img->r.w, img->r.h, fu->sx1, fu->sy1);
        if (!fuse_gi) {     // the separate call it stands for
            int rc0 = artgpu_get_image(ctx, dm, fu->sx1, fu->sy1, fu->mul, fu->do_clip, fu->cam_to_work, img);
            if (rc0) return rc0;
        }
    }
    if (fu && (fu->demosaiced || fu->exposure_enabled)) {
        // one level down with what is left to fuse; the exposure that could not be fused follows as its own call
        struct Restore { artgpu_ctx *c; ~Restore() { c->fuse_gi.on = 0; c->fuse_exp_on = 0; c->tail_exp_on = 0; } } restore{ctx};
        if (fuse_gi) {
            GetImageFuse &g = ctx->fuse_gi;
            g.on = 1;
            g.src[0] = fu->demosaiced->r.p; g.src[1] = fu->demosaiced->g.p; g.src[2] = fu->demosaiced->b.p;
            g.stride = (size_t)(fu->demosaiced->r.row_stride_bytes / 4);
            g.sx1 = fu->sx1; g.sy1 = fu->sy1;
            for (int k = 0; k < 3; ++k) g.mul[k] = fu->mul[k];
            g.do_clip = fu->do_clip ? 1 : 0; g.has_mat = fu->cam_to_work ? 1 : 0;
            for (int k = 0; k < 9; ++k) g.mat[k] = fu->cam_to_work ? fu->cam_to_work[k] : 0.0;
        }
        if (fuse_exp) { ctx->fuse_exp_on = 1; ctx->fuse_exp_scale = fu->exp_scale; ctx->fuse_exp_black = fu->black; }
        if (tail_exp) { ctx->tail_exp_on = 1; ctx->tail_exp_scale = fu->exp_scale; ctx->tail_exp_black = fu->black; }
        int rc0 = artgpu_improc_denoise_fused(ctx, img, nullptr, p, ws, iws, ecomp, scale, calclum_mat, noise_c_curve, flags);
        if (rc0) return rc0;
        if (fu->exposure_enabled && !fuse_exp && !tail_exp) return artgpu_exposure(ctx, img, fu->exp_scale, fu->black);
        return ARTGPU_OK;
    }
    if (!img->r.on_device) {
        // host planes: stage once, run the whole tool on the staged copy, copy back
        DevRGB d;
        int rc0 = bind_rgb(ctx, img, 4, true, &d, "improc_denoise");
        if (rc0) return rc0;
        artgpu_rgb dv;
        artgpu_plane *pl[3] = {&dv.r, &dv.g, &dv.b};
        for (int k = 0; k < 3; ++k) { pl[k]->p = d.p[k]; pl[k]->w = d.w; pl[k]->h = d.h; pl[k]->row_stride_bytes = (int64_t)d.stride * 4; pl[k]->on_device = 1; }
        if ((rc0 = artgpu_improc_denoise(ctx, &dv, p, ws, iws, ecomp, scale, calclum_mat, noise_c_curve, flags))) return rc0;
        return unbind_rgb(ctx, img, &d);
    }
    float wsf[9];
    for (int k = 0; k < 9; ++k) wsf[k] = (float)ws[k];
    int rc;
    // adjust_params (ipdenoise.cc:35-63): a preview at scale > 1 sees averaged-down noise, so the strengths shrink with it
    artgpu_denoise_tool_params adj = *p;
    if (scale > 1.0) {
        const auto c = [](double x, double f) -> double {
            const int sgn = (0.0 < x) - (x < 0.0);
            const double y = std::max(0.0, std::min(std::abs(x) / 100.0, 1.0));
            return sgn * (y * (y * f) + (1.0 - y) * y) * 100.0;      // intp(y, y*f, y)
        };
        const double scale_factor = 1.0 / scale;
        const double noise_factor_c = std::pow(scale_factor, 0.46);
        const double noise_factor_l = std::pow(scale_factor, 0.62) * scale_factor;
        adj.dn.luminance = c(adj.dn.luminance, noise_factor_l);
        adj.dn.luminance_detail *= (1.0 + std::pow(1.0 - scale_factor, 2.2));
        adj.dn.chrominance = c(adj.dn.chrominance, noise_factor_c);
        adj.dn.chrominance_red_green = c(adj.dn.chrominance_red_green, noise_factor_c);
        adj.dn.chrominance_blue_yellow = c(adj.dn.chrominance_blue_yellow, noise_factor_c);
    }
    p = &adj;
    artgpu_plane ccalc = {}, *ccalc_p = nullptr;
    if (noise_c_curve) {
        float sum = 0.f;
        for (int i = 0; i < 501; ++i) sum += noise_c_curve[i];      // NoiseCurve::getSum (ipdenoise.cc:698)
        if (sum > 5.f) {                                            // useNoiseCCurve, FTblockDN.cc:1672
            float *pl[3] = {img->r.p, img->g.p, img->b.p}, *map;
            if (img->g.row_stride_bytes != img->r.row_stride_bytes || img->b.row_stride_bytes != img->r.row_stride_bytes)
                return fail(ctx, ARTGPU_EINVAL, "improc_denoise: planes must share one row stride");
            if ((rc = chroma_map_dev(ctx, pl, (size_t)(img->r.row_stride_bytes / 4), img->r.w, img->r.h, calclum_mat, ws, noise_c_curve, &map))) return rc;
            ccalc.p = map; ccalc.w = (img->r.w + 1) / 2; ccalc.h = (img->r.h + 1) / 2; ccalc.row_stride_bytes = (int64_t)ccalc.w * 4; ccalc.on_device = 1;
            ccalc_p = &ccalc;
        }
    }
    // expcomp(+ecomp) / expcomp(-ecomp) (ipdenoise.cc:1161-1163,1181-1184) are fused into RGB_denoise's first and last pixel
    // passes when RGB_denoise will actually run them (same operations on the same values, two fewer passes over the image)
    const bool dn_runs = !(p->dn.luminance == 0 && p->dn.chrominance == 0 && !ccalc_p);
    const bool fuse_pre = ecomp > 0 && dn_runs, fuse_post = fuse_pre && !p->smoothing_enabled;
    if (ecomp > 0 && !fuse_pre) { if ((rc = artgpu_exposure(ctx, img, (float)std::pow(2.0, ecomp), 0.f))) return rc; }
    ctx->fuse_pre = fuse_pre ? (float)std::pow(2.0, ecomp) : 0.f;
    ctx->fuse_post = fuse_post ? (float)std::pow(2.0, -ecomp) : 0.f;
    float iwsf[9];
    if (iws) for (int k = 0; k < 9; ++k) iwsf[k] = (float)iws[k];
    ctx->ccalc_nonneg = ccalc_p ? 1 : 0;
    rc = artgpu_rgb_denoise(ctx, img, &p->dn, wsf, iws ? iwsf : nullptr, 0.0, scale, ccalc_p, flags, nullptr, nullptr);
    ctx->ccalc_nonneg = 0;
    ctx->fuse_pre = ctx->fuse_post = 0.f;
    if (rc) return rc;
    // the exposure steps behind the tool's last stage -- its own expcomp(-ecomp) (L1181-1184) where yuv2rgb could not take it, and the STAGE_1
    // exposure of artgpu_improc_denoise_fused -- ride on the last pixel pass there is: setMode(RGB) behind NL-means, or one exposure pass
    int chain_n = 0;
    float chain_scale[2], chain_black[2];
    if (ecomp > 0 && !fuse_post) { chain_scale[chain_n] = (float)std::pow(2.0, -ecomp); chain_black[chain_n] = 0.f; ++chain_n; }
    if (ctx->tail_exp_on) { chain_scale[chain_n] = ctx->tail_exp_scale; chain_black[chain_n] = ctx->tail_exp_black; ++chain_n; }
    bool chained = false;
    if (p->smoothing_enabled) {
        if ((rc = artgpu_denoise_guided_smoothing(ctx, img, ws, p->guided_chroma_radius, scale))) return rc;
        if (p->nl_strength) {
            PixArgs a = {};
            float *pl[3] = {img->r.p, img->g.p, img->b.p};
            for (int k = 0; k < 3; ++k) { a.dst[k] = pl[k]; a.mul[k] = wsf[3 + k]; }
            a.dst_stride = (size_t)(img->r.row_stride_bytes / 4); a.w = img->r.w; a.h = img->r.h;
            a.do_clip = 0;
            HIPCHK(ctx, launch_yuv_mode(a, ctx->stream));
            if ((rc = artgpu_nlmeans(ctx, &img->g, 65535.f, p->nl_strength, p->nl_detail, (float)scale))) return rc;
            a.do_clip = 1;
            a.chain_n = chain_n;
            for (int k = 0; k < chain_n; ++k) { a.chain_scale[k] = chain_scale[k]; a.chain_black[k] = chain_black[k]; }
            HIPCHK(ctx, launch_yuv_mode(a, ctx->stream));
            chained = true;
        }
    }
    if (chain_n && !chained) {
        PixArgs a = {};
        float *pl[3] = {img->r.p, img->g.p, img->b.p};
        for (int k = 0; k < 3; ++k) a.dst[k] = pl[k];
        a.dst_stride = (size_t)(img->r.row_stride_bytes / 4); a.w = img->r.w; a.h = img->r.h;
        a.exp_scale = chain_scale[0]; a.black = chain_black[0];
        a.chain_n = chain_n - 1;
        if (chain_n > 1) { a.chain_scale[0] = chain_scale[1]; a.chain_black[0] = chain_black[1]; }
        HIPCHK(ctx, launch_exposure(a, ctx->stream));
    }
    return ARTGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// N4: channel mixer, RGB curves
// ---------------------------------------------------------------------------------------------
int artgpu_channel_mixer(artgpu_ctx *ctx, artgpu_rgb *image, const float m[9])
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!image || !m) return fail(ctx, ARTGPU_EINVAL, "channel_mixer: null argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevRGB d;
    int rc = bind_rgb(ctx, image, 4, true, &d, "channel_mixer");
    if (rc) return rc;
    MixArgs a = {};
    for (int k = 0; k < 3; ++k) a.dst[k] = d.p[k];
    a.stride = d.stride; a.w = d.w; a.h = d.h;
    for (int k = 0; k < 9; ++k) a.m[k] = m[k];
    HIPCHK(ctx, launch_channel_mixer(a, ctx->stream));
    return unbind_rgb(ctx, image, &d);
}

int artgpu_rgb2out_matrix(artgpu_ctx *ctx, const artgpu_rgb *src, artgpu_rgb *dst, const float matrix[9], int trc_linear,
                          const float *lut, int lutsz)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!src || !dst || !matrix) return fail(ctx, ARTGPU_EINVAL, "rgb2out_matrix: null argument");
    if (!trc_linear && (!lut || lutsz < 2)) return fail(ctx, ARTGPU_EINVAL, "rgb2out_matrix: a non-linear TRC needs its LUT");
    if (lut && (lutsz < 2 || lutsz > 65536)) return fail(ctx, ARTGPU_EINVAL, "rgb2out_matrix: lutsz %d", lutsz);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevRGB s, d;
    int rc = bind_rgb(ctx, src, 1, true, &s, "rgb2out_matrix(src)");
    if (rc) return rc;
    if ((rc = bind_rgb(ctx, dst, 4, false, &d, "rgb2out_matrix(dst)"))) return rc;
    if (s.w != d.w || s.h != d.h) return fail(ctx, ARTGPU_EINVAL, "rgb2out_matrix: size mismatch");
    float *tab;
    if ((rc = pool_get(ctx, P_PIPE_R, (65536 + 16) * 4, &tab))) return rc;
    OutArgs a = {};
    for (int k = 0; k < 3; ++k) { a.src[k] = s.p[k]; a.dst[k] = d.p[k]; }
    a.src_stride = s.stride; a.dst_stride = d.stride; a.w = s.w; a.h = s.h;
    for (int k = 0; k < 9; ++k) a.m[k] = matrix[k];
    a.linear = trc_linear ? 1 : 0;
    a.unsupported = reinterpret_cast<int *>(tab + 65536);
    HIPCHK(ctx, hipMemsetAsync(a.unsupported, 0, 4, ctx->stream));
    if (lut) {
        if ((rc = h2d_table(ctx, tab, lut, (size_t)lutsz * 4))) return rc;
        a.lut = tab; a.lutsz = lutsz;
    }
    HIPCHK(ctx, launch_rgb2out_matrix(a, ctx->stream));
    int bad = 0;
    HIPCHK(ctx, hipMemcpyAsync(&bad, a.unsupported, 4, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = unbind_rgb(ctx, dst, &d))) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (bad) return fail(ctx, ARTGPU_EUNSUPPORTED, "rgb2out_matrix: %d channel values above 1 need ARTOutputProfile::eval (lcms2/libm) on the host", bad);
    return ARTGPU_OK;
}

int artgpu_get_scanlines(artgpu_ctx *ctx, const artgpu_rgb *img, int bps, int is_float, void *dst, int64_t dst_row_stride_bytes, int dst_on_device)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!img || !dst) return fail(ctx, ARTGPU_EINVAL, "get_scanlines: null argument");
    const bool okfmt = is_float ? (bps == 16 || bps == 32) : (bps == 8 || bps == 16);
    if (!okfmt) return fail(ctx, ARTGPU_EINVAL, "get_scanlines: bps %d / is_float %d", bps, is_float);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevRGB s;
    int rc = bind_rgb(ctx, img, 4, true, &s, "get_scanlines");
    if (rc) return rc;
    const size_t rowb = (size_t)s.w * 3 * (bps / 8);
    if (dst_row_stride_bytes < (int64_t)rowb) return fail(ctx, ARTGPU_EINVAL, "get_scanlines: row stride %lld < %zu", (long long)dst_row_stride_bytes, rowb);
    OutArgs a = {};
    for (int k = 0; k < 3; ++k) a.src[k] = s.p[k];
    a.src_stride = s.stride; a.w = s.w; a.h = s.h; a.bps = bps; a.is_float = is_float ? 1 : 0;
    float *stage = nullptr;
    if (dst_on_device) { a.out = static_cast<unsigned char *>(dst); a.out_stride_bytes = (size_t)dst_row_stride_bytes; }
    else {
        if ((rc = pool_get(ctx, P_TMP, rowb * s.h + 16, &stage))) return rc;
        a.out = reinterpret_cast<unsigned char *>(stage); a.out_stride_bytes = rowb;
    }
    HIPCHK(ctx, launch_scanlines(a, ctx->stream));
    if (!dst_on_device) {
        HIPCHK(ctx, hipMemcpy2DAsync(dst, (size_t)dst_row_stride_bytes, stage, rowb, rowb, s.h, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return ARTGPU_OK;
}

int artgpu_saturation_vibrance(artgpu_ctx *ctx, artgpu_rgb *image, int saturation, int vibrance, const double ws[9])
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!image || !ws) return fail(ctx, ARTGPU_EINVAL, "saturation_vibrance: null argument");
    if (!saturation && !vibrance) return ARTGPU_OK;            // ipsaturation.cc:45-46
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevRGB d;
    int rc = bind_rgb(ctx, image, 4, true, &d, "saturation_vibrance");
    if (rc) return rc;
    SatArgs a = {};
    for (int k = 0; k < 3; ++k) { a.dst[k] = d.p[k]; a.ws1[k] = ws[3 + k]; }
    a.stride = d.stride; a.w = d.w; a.h = d.h;
    a.saturation = 1.f + saturation / 100.f;
    a.vibrance = 1.f - vibrance / 1000.f;
    a.vib = vibrance ? 1 : 0;
    HIPCHK(ctx, launch_saturation_vibrance(a, ctx->stream));
    return unbind_rgb(ctx, image, &d);
}

int artgpu_rgb_curves(artgpu_ctx *ctx, artgpu_rgb *image, const float *rcurve, const float *gcurve, const float *bcurve)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!image) return fail(ctx, ARTGPU_EINVAL, "rgb_curves: null argument");
    if (!rcurve && !gcurve && !bcurve) return ARTGPU_OK;    // all identity: the reference skips the loop (iprgbcurves.cc:110)
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevRGB d;
    int rc = bind_rgb(ctx, image, 4, true, &d, "rgb_curves");
    if (rc) return rc;
    // The tables live in a pool slot of their own: the host-side cache below describes what the slot holds, so nothing else may write it
    // (P_PIPE_R, where they used to be, is also labAdjustments' curves, rgb2out's TRC table and the pipeline's red plane).
    float *tabs;
    if (ctx->pool_bytes[P_RGBCURVES] < (size_t)3 * 65536 * 4)
        for (int k = 0; k < 3; ++k) ctx->rgbcurve_host[k].clear();           // a fresh allocation holds nothing
    if ((rc = pool_get(ctx, P_RGBCURVES, 3 * 65536 * 4, &tabs))) return rc;
    const float *host[3] = {rcurve, gcurve, bcurve};
    MixArgs a = {};
    for (int k = 0; k < 3; ++k) {
        a.dst[k] = d.p[k];
        if (host[k]) {
            // one rule for every look-up table that crosses the boundary (artgpu.h "Host look-up tables"): the call keeps its own copy and
            // the caller's array is free when the call returns; the same table call after call is uploaded once
            std::vector<float> &own = ctx->rgbcurve_host[k];
            if (own.size() != 65536 || std::memcmp(own.data(), host[k], 65536 * 4) != 0) {
                own.assign(host[k], host[k] + 65536);
                hipError_t e = hipMemcpyAsync(tabs + (size_t)k * 65536, own.data(), 65536 * 4, hipMemcpyHostToDevice, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess) { own.clear(); return fail(ctx, ARTGPU_EHIP, "rgb_curves: curve upload failed: %s", hipGetErrorString(e)); }
            }
            a.lut[k] = tabs + (size_t)k * 65536;
        }
    }
    a.stride = d.stride; a.w = d.w; a.h = d.h;
    HIPCHK(ctx, launch_rgb_curves(a, ctx->stream));
    rc = unbind_rgb(ctx, image, &d);
    if (rc) return rc;
    return ARTGPU_OK;      // (host LUT lifetime: see "Host look-up tables" in artgpu.h)
}

// ---------------------------------------------------------------------------------------------
// raw pre-stage (N2): copyOriginalPixels + scaleColors
// ---------------------------------------------------------------------------------------------
int artgpu_scale_colors(artgpu_ctx *ctx, const void *src, int32_t w, int32_t h, int64_t src_row_stride_bytes, int32_t src_is_u16,
                        int32_t src_on_device, uint32_t filters, const int32_t *xtrans, const float cblacksom[4],
                        const float scale_mul[4], artgpu_plane *dst, float chmax[4])
{
    if (!ctx) return ARTGPU_EINVAL;
    const int esz = src_is_u16 ? 2 : 4;
    if (!src || !dst || !plane_ok(dst) || !cblacksom || !scale_mul || w <= 0 || h <= 0 || dst->w != w || dst->h != h ||
        src_row_stride_bytes < (int64_t)w * esz || src_row_stride_bytes % esz)
        return fail(ctx, ARTGPU_EINVAL, "scale_colors: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    ScaleArgs a = {};
    a.w = w; a.h = h; a.src_u16 = src_is_u16 ? 1 : 0;
    // source: device pointer as is, host buffer staged (2 or 4 bytes per pixel over PCIe)
    if (src_on_device) { a.src = src; a.src_stride = (size_t)(src_row_stride_bytes / esz); }
    else {
        float *st;
        if ((rc = pool_get(ctx, P_PIPE_B, (size_t)w * h * esz, &st))) return rc;
        HIPCHK(ctx, hipMemcpy2DAsync(st, (size_t)w * esz, src, (size_t)src_row_stride_bytes, (size_t)w * esz, h, hipMemcpyHostToDevice, ctx->stream));
        a.src = st; a.src_stride = w;
    }
    float *out;
    if (dst->on_device) { out = dst->p; a.dst_stride = (size_t)(dst->row_stride_bytes / 4); }
    else { if ((rc = pool_get(ctx, P_PIPE_G, (size_t)w * h * 4, &out))) return rc; a.dst_stride = w; }
    a.dst = out;
    a.bayer = xtrans ? 0 : 1;
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
            int v;
            if (xtrans) v = xtrans[r * 6 + c];
            else v = (filters >> ((((r << 1) & 14) + (c & 1)) << 1)) & 3;      // RawImage::FC, period 2 tiles the 6x6 map
            if (v < 0 || v > 2) return fail(ctx, ARTGPU_EUNSUPPORTED, "scale_colors: three-colour CFAs only");
            a.cfa[r * 6 + c] = v;
        }
    for (int k = 0; k < 4; ++k) { a.cblacksom[k] = cblacksom[k]; a.scale_mul[k] = scale_mul[k]; }
    float *mx;
    if ((rc = pool_get(ctx, P_MAD, 3 * 32 * 4, &mx))) return rc;
    a.chmax_bits = reinterpret_cast<int *>(mx);
    HIPCHK(ctx, hipMemsetAsync(mx, 0, 4 * sizeof(int), ctx->stream));
    HIPCHK(ctx, launch_scale_colors(a, ctx->stream));
    if (!dst->on_device)
        HIPCHK(ctx, hipMemcpy2DAsync(dst->p, (size_t)dst->row_stride_bytes, out, (size_t)w * 4, (size_t)w * 4, h, hipMemcpyDeviceToHost, ctx->stream));
    if (chmax) {
        float host[4];
        HIPCHK(ctx, hipMemcpyAsync(host, mx, 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        chmax[0] = host[0]; chmax[1] = host[1]; chmax[2] = host[2]; chmax[3] = host[1];
    } else if (!dst->on_device) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return ARTGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// one frame / one batch share through the whole path
// ---------------------------------------------------------------------------------------------
int artgpu_pipeline_run(artgpu_ctx *ctx, const artgpu_plane *raw, const artgpu_pipeline_params *p, artgpu_rgb *out)
{
    StageScope scope_(ctx, "ImageProcessor (stage_init .. stage_finish)");
    if (!ctx) return ARTGPU_EINVAL;
    if (!plane_ok(raw) || !p || !out) return fail(ctx, ARTGPU_EINVAL, "pipeline_run: null/bad argument");
    const int W = raw->w, H = raw->h, b = p->border;
    if (b < 0 || W - 2 * b < 8 || H - 2 * b < 8) return fail(ctx, ARTGPU_EINVAL, "pipeline_run: border %d leaves no image", b);
    if (out->r.w != W - 2 * b || out->r.h != H - 2 * b) return fail(ctx, ARTGPU_EINVAL, "pipeline_run: output must be %dx%d", W - 2 * b, H - 2 * b);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    // demosaiced planes live in the context pool (never leave the device)
    float *pl[3];
    for (int k = 0; k < 3; ++k)
        if ((rc = pool_get(ctx, P_PIPE_R + k, (size_t)W * H * 4, &pl[k]))) return rc;
    artgpu_rgb dem;
    artgpu_plane *dp[3] = {&dem.r, &dem.g, &dem.b};
    for (int k = 0; k < 3; ++k) { dp[k]->p = pl[k]; dp[k]->w = W; dp[k]->h = H; dp[k]->row_stride_bytes = (int64_t)W * 4; dp[k]->on_device = 1; }
    if (p->sensor == 0) rc = artgpu_demosaic_bayer(ctx, p->bayer_method, raw, p->filters, p->initial_gain, b, &dem);
    else rc = artgpu_demosaic_xtrans(ctx, p->xtrans_passes, p->xtrans_passes > 1 ? 1 : 0, raw, p->xtrans, p->rgb_cam, &dem);
    if (rc) return rc;
    // the image the remaining stages work on: the caller's planes if they are on the device, else a staged copy
    DevRGB d;
    if ((rc = bind_rgb(ctx, out, 4, false, &d, "pipeline_run(out)"))) return rc;
    artgpu_rgb img;
    artgpu_plane *ip[3] = {&img.r, &img.g, &img.b};
    for (int k = 0; k < 3; ++k) { ip[k]->p = d.p[k]; ip[k]->w = d.w; ip[k]->h = d.h; ip[k]->row_stride_bytes = (int64_t)d.stride * 4; ip[k]->on_device = 1; }
    artgpu_denoise_tool_params dnp = p->denoise;
    if (p->denoise_enabled && dnp.dn.chrominance_method == 1) {
        // ipf.denoiseComputeParams(imgsrc, currWB, dnstore, params.denoise) before getImage (simpleprocess.cc:254-256); a fresh store per frame
        static const double ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        artgpu_denoise_info_store store = {};
        if ((rc = artgpu_denoise_compute_params(ctx, &dem, b, p->mul, p->do_clip, p->has_cam_to_work ? p->cam_to_work : ident, p->ws,
                                                p->chrominance_auto_factor != 0.0 ? p->chrominance_auto_factor : 1.0, &store, &dnp.dn)))
            return rc;
    }
    if (p->denoise_enabled) {
        // getImage + convertColorSpace in front of the tool and the STAGE_1 exposure behind it live in the tool's own pixel passes where
        // its parameters allow (artgpu_improc_denoise_fused: otherwise they run as the separate calls)
        static const double curve_points[9] = {1 /*FCT_MinMaxCPoints*/, 0.05, 0.50, 0.35, 0.35, 0.35, 0.05, 0.35, 0.35};   // ipdenoise.cc:1139-1149
        float curve[501];
        (void)noise_curve_lut(curve_points, 9, curve);
        const double ecomp = p->exposure_enabled ? p->expcomp : 0.0;       // ipdenoise.cc:1155
        artgpu_denoise_fusion fu = {};
        fu.demosaiced = &dem; fu.sx1 = b; fu.sy1 = b;
        for (int k = 0; k < 3; ++k) fu.mul[k] = p->mul[k];
        fu.do_clip = p->do_clip; fu.cam_to_work = p->has_cam_to_work ? p->cam_to_work : nullptr;
        fu.exposure_enabled = p->exposure_enabled ? 1 : 0;
        fu.exp_scale = (float)std::pow(2.0, p->expcomp); fu.black = (float)(p->black * 2000.0);
        if ((rc = artgpu_improc_denoise_fused(ctx, &img, &fu, &dnp, p->ws, p->iws, ecomp, p->scale > 0 ? p->scale : 1.0,
                                              p->has_cam_to_work ? p->cam_to_work : nullptr, curve, 0u)))
            return rc;
    } else {
        if ((rc = artgpu_get_image(ctx, &dem, b, b, p->mul, p->do_clip, p->has_cam_to_work ? p->cam_to_work : nullptr, &img))) return rc;
        if (p->exposure_enabled)
            if ((rc = artgpu_exposure(ctx, &img, (float)std::pow(2.0, p->expcomp), (float)(p->black * 2000.0)))) return rc;
    }
    if (p->tone_enabled) {
        if (p->tone_mode == ARTGPU_TONE_NEUTRAL) {
            artgpu_neutral_state st;
            for (int k = 0; k < 9; ++k) { st.ws[k] = p->ws[k]; st.iws[k] = p->iws[k]; st.to_out[k] = p->to_out[k]; st.to_work[k] = p->to_work[k]; }
            rc = artgpu_tone_curve_neutral(ctx, &img, p->tone_lut, p->white_point, &st);
        } else {
            rc = artgpu_tone_curve(ctx, &img, p->tone_mode, p->tone_lut, p->white_point, 1);
        }
        if (rc) return rc;
    }
    return unbind_rgb(ctx, out, &d);
}

int artgpu_set_batch_lanes(artgpu_ctx *ctx, int lanes)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (lanes < 1 || lanes > 8) return fail(ctx, ARTGPU_EINVAL, "set_batch_lanes: 1..8");
    ctx->batch_lanes = lanes;
    return ARTGPU_OK;
}

namespace {
// The scratch of a context only grows (about 5 GB for a 45 MP frame through the whole tool).  A batch that goes on with much smaller frames
// would keep the large frame's pools for nothing: when this frame AND the lane's next one have less than half the pixels the pools were
// grown for, the lane gives them back first (artgpu_trim_scratch: a stream drain and a few hipFree, paid once per such transition; a
// batch that alternates sizes keeps its pools).  Round-5 review, weak point 10.  `nxt` < 0: the lane's last frame has no successor to go
// by -- it keeps the pools, a caller that is done calls artgpu_trim_scratch.
int batch_settle_scratch(artgpu_ctx *c, long long px, long long nxt)
{
    if (nxt < 0) nxt = c->batch_px;
    if (c->batch_px && 2 * px < c->batch_px && 2 * nxt < c->batch_px) {
        std::vector<artgpu_ctx *> keep;
        keep.swap(c->lanes);                       // (only this context's own pools: its lanes decide for themselves)
        const int rc = artgpu_trim_scratch(c);
        keep.swap(c->lanes);
        if (rc) return rc;
        c->batch_px = 0;
    }
    if (px > c->batch_px) c->batch_px = px;
    return ARTGPU_OK;
}

// Frames are independent: lane k (its own context, stream and host thread) takes frames k, k+L, ...  The kernels of one frame
// are a mix of latency-bound (AMaZE) and bandwidth-bound (wavelet passes) work, so frames in flight on different streams fill
// each other's gaps (+11 % throughput with three lanes at 45 MP, scripts/overlap_time.py).
int batch_prepare_lanes(artgpu_ctx *ctx, int L)
{
    HIPCHK(ctx, hipSetDevice(ctx->device));
    while ((int)ctx->lanes.size() < L - 1) {
        artgpu_ctx *peer = nullptr;
        int rc = artgpu_create(ctx->device, &peer);
        if (rc) return fail(ctx, rc, "batch_run: cannot create lane %d", (int)ctx->lanes.size() + 1);
        if (hipStreamCreateWithFlags(&peer->stream, hipStreamNonBlocking) != hipSuccess) { (void)artgpu_destroy(peer); return fail(ctx, ARTGPU_EHIP, "batch_run: stream"); }
        peer->owns_stream = true;
        ctx->lanes.push_back(peer);
    }
    // the lanes are this context as far as the caller can tell: its options, curve tail and progress listener apply to every frame,
    // whichever lane runs it (copied on every call -- they may change between calls)
    for (artgpu_ctx *peer : ctx->lanes) {
        peer->curve_tail_kind = ctx->curve_tail_kind; peer->curve_tail_y = ctx->curve_tail_y; peer->curve_tail_pc = ctx->curve_tail_pc;
        peer->opt_amaze_path = ctx->opt_amaze_path; peer->opt_amaze_split = ctx->opt_amaze_split; peer->opt_amaze_overlap = ctx->opt_amaze_overlap; peer->opt_amaze_grid = ctx->opt_amaze_grid;
        peer->opt_amaze_zero_mask = ctx->opt_amaze_zero_mask; peer->opt_amaze_zero_frame = ctx->opt_amaze_zero_frame; peer->opt_amaze_poison = ctx->opt_amaze_poison;
        peer->opt_rcd_rows = ctx->opt_rcd_rows; peer->opt_roctx = ctx->opt_roctx; peer->opt_lut_lds = ctx->opt_lut_lds; peer->opt_dn_streams = ctx->opt_dn_streams; peer->opt_dn_fused = ctx->opt_dn_fused;
        peer->opt_dn_wait_ms = ctx->opt_dn_wait_ms; peer->opt_dn_debug_stall = ctx->opt_dn_debug_stall; peer->opt_io_direct = ctx->opt_io_direct;
        peer->progress_fn = ctx->progress_fn; peer->progress_user = ctx->progress_user;
        peer->frames_in_flight = L;
    }
    ctx->frames_in_flight = L;
    return ARTGPU_OK;
}
struct InFlightReset { artgpu_ctx *c; ~InFlightReset() { c->frames_in_flight = 1; for (artgpu_ctx *p : c->lanes) p->frames_in_flight = 1; } };

// run `work(lane context, lane index)` on L lanes (lane 0 on the calling thread) and gather the status codes
extern "C++" {
template <typename F>
int batch_on_lanes(artgpu_ctx *ctx, int L, F work)
{
    std::vector<int> rcs(L, ARTGPU_OK);
    // (the lanes' contexts are looked up HERE: lane 0 may take ctx->lanes aside for a moment while it trims its own pools -- batch_settle_scratch --,
    // and a thread that starts late must not index the vector then)
    std::vector<artgpu_ctx *> cs(L, ctx);
    for (int k = 1; k < L; ++k) cs[k] = ctx->lanes[k - 1];
    std::vector<std::thread> threads;
    for (int k = 1; k < L; ++k) threads.emplace_back([&rcs, &work, &cs, k]() { rcs[k] = work(cs[k], k); });
    rcs[0] = work(ctx, 0);
    for (std::thread &t : threads) t.join();
    for (int k = 0; k < L; ++k)
        if (rcs[k]) return fail(ctx, rcs[k], "batch_run: lane %d: %s", k, k == 0 ? ctx->err.c_str() : cs[k]->err.c_str());
    return ARTGPU_OK;
}
} // extern "C++"
} // namespace

int artgpu_batch_run(artgpu_ctx *ctx, int nframes, const artgpu_plane *raws, const artgpu_pipeline_params *params, artgpu_rgb *outs)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (nframes < 0 || (nframes && (!raws || !params || !outs))) return fail(ctx, ARTGPU_EINVAL, "batch_run: null argument");
    const int L = std::min(ctx->batch_lanes, nframes);
    auto px_of = [&](int f) { return (long long)raws[f].w * raws[f].h; };
    if (L <= 1) {
        for (int f = 0; f < nframes; ++f) {
            int rc = batch_settle_scratch(ctx, px_of(f), f + 1 < nframes ? px_of(f + 1) : -1);
            if (!rc) rc = artgpu_pipeline_run(ctx, &raws[f], params, &outs[f]);
            if (rc) return rc;
        }
        return ARTGPU_OK;
    }
    int rc = batch_prepare_lanes(ctx, L);
    if (rc) return rc;
    InFlightReset in_flight_reset{ctx};
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));      // inputs the caller produced on this context's stream
    return batch_on_lanes(ctx, L, [&](artgpu_ctx *c, int k) -> int {
        for (int f = k; f < nframes; f += L) {
            int rc2 = batch_settle_scratch(c, px_of(f), f + L < nframes ? px_of(f + L) : -1);
            if (!rc2) rc2 = artgpu_pipeline_run(c, &raws[f], params, &outs[f]);
            if (rc2) return rc2;
        }
        if (k > 0 && hipStreamSynchronize(c->stream) != hipSuccess) return fail(c, ARTGPU_EHIP, "batch_run: lane %d: stream", k);
        return ARTGPU_OK;
    });
}

// ---------------------------------------------------------------------------------------------
// the batch between the decoder's and the writers' formats, copies beside the kernels
// ---------------------------------------------------------------------------------------------
namespace {
int io_setup(artgpu_ctx *c, int nfr)
{
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->io_up) HIPCHK(c, hipStreamCreateWithFlags(&c->io_up, hipStreamNonBlocking));
    if (!c->io_down) HIPCHK(c, hipStreamCreateWithFlags(&c->io_down, hipStreamNonBlocking));
    for (int k = 0; k < 8; ++k)
        if (!c->io_ev[k]) HIPCHK(c, hipEventCreateWithFlags(&c->io_ev[k], hipEventDisableTiming));
    if (c->io_host_cap < nfr) {
        if (c->io_host) { HIPCHK(c, hipHostFree(c->io_host)); c->io_host = nullptr; c->io_host_cap = 0; }
        const int cap = nfr < 16 ? 16 : nfr;
        if (hipHostMalloc(reinterpret_cast<void **>(&c->io_host), (size_t)cap * 8 * sizeof(int), hipHostMallocDefault) != hipSuccess) {
            c->io_host = nullptr;
            return fail(c, ARTGPU_ENOMEM, "batch_run_io: pinned flag words for %d frames", cap);
        }
        c->io_host_cap = cap;
    }
    std::memset(c->io_host, 0, (size_t)c->io_host_cap * 8 * sizeof(int));
    return ARTGPU_OK;
}

// pinned (hipHostMalloc / hipHostRegister) host memory?  Its device address if so.  Pageable memory is "invalid value" to the query: not an error here.
void *pinned_device_address(const void *p)
{
    hipPointerAttribute_t at = {};
    if (hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) return at.devicePointer;
    (void)hipGetLastError();
    return nullptr;
}

// frame `i` of this lane: everything is queued, nothing is waited for (the staging slots of turn i - 2 are released through events)
int io_frame(artgpu_ctx *c, int i, const artgpu_sensor_frame *in, const artgpu_pipeline_params *p, artgpu_scanline_frame *out)
{
    const int s = i & 1, W = in->w, H = in->h, b = p->border;
    const int esz = in->is_u16 ? 2 : 4;
    if (!in->data || W <= 0 || H <= 0 || in->row_stride_bytes < (int64_t)W * esz || in->row_stride_bytes % esz)
        return fail(c, ARTGPU_EINVAL, "batch_run_io: sensor frame: bad pointer/size/stride");
    if (b < 0 || W - 2 * b < 8 || H - 2 * b < 8) return fail(c, ARTGPU_EINVAL, "batch_run_io: border %d leaves no image", b);
    const int iw = W - 2 * b, ih = H - 2 * b;
    const bool okfmt = out->is_float ? (out->bps == 16 || out->bps == 32) : (out->bps == 8 || out->bps == 16);
    if (!okfmt) return fail(c, ARTGPU_EINVAL, "batch_run_io: bps %d / is_float %d", out->bps, out->is_float);
    const size_t rowb = (size_t)iw * 3 * (out->bps / 8);
    if (!out->scanlines || out->row_stride_bytes < (int64_t)rowb) return fail(c, ARTGPU_EINVAL, "batch_run_io: scanlines: bad pointer / row stride < %zu", rowb);
    if (out->rgb2out_enabled) {
        if (!out->trc_linear && (!out->trc_lut || out->trc_lutsz < 2)) return fail(c, ARTGPU_EINVAL, "batch_run_io: a non-linear TRC needs its LUT");
        if (out->trc_lut && (out->trc_lutsz < 2 || out->trc_lutsz > 65536)) return fail(c, ARTGPU_EINVAL, "batch_run_io: trc_lutsz %d", out->trc_lutsz);
    }
    int rc;
    float *cfa, *img[3], *flags_f;
    if ((rc = pool_get(c, P_IO_CFA, (size_t)W * H * 4, &cfa)) || (rc = pool_get(c, P_IO_FLAGS, 2 * 8 * sizeof(int), &flags_f))) return rc;
    for (int k = 0; k < 3; ++k)
        if ((rc = pool_get(c, P_IO_IMG0 + 3 * s + k, (size_t)iw * ih * 4, &img[k]))) return rc;      // (two sets: the download of turn i reads its set while turn i + 1 is computed)
    int *flags = reinterpret_cast<int *>(flags_f) + 8 * s;
    // Where do the scanlines go?  Pinned host memory is mapped into the device's address space: a few workgroups on the download stream write
    // them there directly.  Anything else (pageable memory, rows that are not 16-byte aligned, option io_direct = 0) is staged and copied.
    // Scanlines for PINNED memory can be written there by a kernel of the download stream (option io_direct = n workgroups) instead of being staged and
    // copied.  Measured on 8192 x 5464 frames, 16-bit scanlines (scripts/pcie_batch.py, batches of 24 - 36 frames, seven runs): the runtime's copy -- itself a
    // kernel, 268 MB in 4.9 ms -- 11.0 ms per frame with one lane, 10.6 - 13.8 with two (two modes: the lanes lock into step with the copy kernel in
    // about half the runs), 10.7 - 11.3 with three; eight workgroups writing directly 11.7 - 12.6 / 11.2 - 11.8 / 11.7 - 12.9 (they hold eight CUs for
    // 5 ms, and the one-workgroup-per-CU kernels of the next frame wait for a CU or run on fewer: cu_reserve).  So: the copy, except with two lanes.
    const int direct_wgs = c->opt_io_direct >= 0 ? c->opt_io_direct : (c->frames_in_flight == 2 ? 8 : 0);
    // A copy from / to PAGEABLE memory blocks the calling thread until it is done (the runtime stages it in pieces): nothing is gained by giving it a
    // stream of its own, so it stays on the context's stream like the copies of the four separate entry points; lanes still overlap each other.
    unsigned char *const out_pinned = out->on_device ? nullptr : static_cast<unsigned char *>(pinned_device_address(out->scanlines));
    const hipStream_t s_up = (in->on_device || pinned_device_address(in->data)) ? c->io_up : c->stream;
    const hipStream_t s_down = out_pinned ? c->io_down : c->stream;
    unsigned char *host_dev = nullptr;
    if (out_pinned && direct_wgs > 0 && (reinterpret_cast<uintptr_t>(out->scanlines) & 15) == 0 && (out->row_stride_bytes & 15) == 0) host_dev = out_pinned;
    c->cu_reserve = host_dev ? direct_wgs * (c->frames_in_flight > 1 ? 2 : 1) : 0;      // (this lane's download and a neighbour's; reset when the lane is done)
    // the working image of this slot is free again once the download of turn i - 2 has read it
    if (i >= 2 && !out->on_device) HIPCHK(c, hipStreamWaitEvent(c->stream, c->io_ev[6 + s], 0));

    // ---- upload (io_up) -> copyOriginalPixels + scaleColors (stream)
    ScaleArgs a = {};
    a.w = W; a.h = H; a.src_u16 = in->is_u16 ? 1 : 0;
    if (in->on_device) { a.src = in->data; a.src_stride = (size_t)(in->row_stride_bytes / esz); }
    else {
        float *st;
        if ((rc = pool_get(c, P_IO_IN0 + s, (size_t)W * H * esz, &st))) return rc;
        if (i >= 2) HIPCHK(c, hipStreamWaitEvent(s_up, c->io_ev[2 + s], 0));       // the kernel that read this slot two turns ago
        // (always the pitched form: a LINEAR copy from or to pageable memory makes the runtime pin the caller's pages for the occasion, 25 ms for a frame's scanlines)
        HIPCHK(c, hipMemcpy2DAsync(st, (size_t)W * esz, in->data, (size_t)in->row_stride_bytes, (size_t)W * esz, H, hipMemcpyHostToDevice, s_up));
        HIPCHK(c, hipEventRecord(c->io_ev[s], s_up));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->io_ev[s], 0));
        a.src = st; a.src_stride = W;
    }
    a.dst = cfa; a.dst_stride = W;
    a.bayer = p->sensor == 0 ? 1 : 0;
    for (int r = 0; r < 6; ++r)
        for (int col = 0; col < 6; ++col) {
            int v;
            if (p->sensor != 0) v = p->xtrans[r * 6 + col];
            else v = (p->filters >> ((((r << 1) & 14) + (col & 1)) << 1)) & 3;
            if (v < 0 || v > 2) return fail(c, ARTGPU_EUNSUPPORTED, "batch_run_io: three-colour CFAs only");
            a.cfa[r * 6 + col] = v;
        }
    for (int k = 0; k < 4; ++k) { a.cblacksom[k] = in->cblacksom[k]; a.scale_mul[k] = in->scale_mul[k]; }
    a.chmax_bits = flags;
    HIPCHK(c, hipMemsetAsync(flags, 0, 8 * sizeof(int), c->stream));
    HIPCHK(c, launch_scale_colors(a, c->stream));
    if (!in->on_device) HIPCHK(c, hipEventRecord(c->io_ev[2 + s], c->stream));

    // ---- the path
    artgpu_plane raw = {cfa, W, H, (int64_t)W * 4, 1};
    artgpu_rgb image;
    artgpu_plane *ip[3] = {&image.r, &image.g, &image.b};
    for (int k = 0; k < 3; ++k) { ip[k]->p = img[k]; ip[k]->w = iw; ip[k]->h = ih; ip[k]->row_stride_bytes = (int64_t)iw * 4; ip[k]->on_device = 1; }
    if ((rc = artgpu_pipeline_run(c, &raw, p, &image))) return rc;

    // ---- rgb2out (matrix + TRC, in place) and the writers' scanlines
    OutArgs o = {};
    for (int k = 0; k < 3; ++k) { o.src[k] = img[k]; o.dst[k] = img[k]; }
    o.src_stride = iw; o.dst_stride = iw; o.w = iw; o.h = ih;
    if (out->rgb2out_enabled) {
        for (int k = 0; k < 9; ++k) o.m[k] = out->out_matrix[k];
        o.linear = out->trc_linear ? 1 : 0;
        o.unsupported = flags + 4;
        if (out->trc_lut) {
            float *tab;
            if ((rc = pool_get(c, P_PIPE_R, (size_t)65536 * 4, &tab)) || (rc = h2d_table(c, tab, out->trc_lut, (size_t)out->trc_lutsz * 4))) return rc;   // (the demosaiced plane in this slot is done with)
            o.lut = tab; o.lutsz = out->trc_lutsz;
        }
        HIPCHK(c, launch_rgb2out_matrix(o, c->stream));
    }
    o.bps = out->bps; o.is_float = out->is_float ? 1 : 0;
    if (host_dev) {
        // flags, then the scanlines on the download stream, straight into the caller's buffer
        HIPCHK(c, hipMemcpyAsync(c->io_host + 8 * i, flags, 8 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipEventRecord(c->io_ev[4 + s], c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->io_down, c->io_ev[4 + s], 0));
        o.out = host_dev; o.out_stride_bytes = (size_t)out->row_stride_bytes;
        HIPCHK(c, launch_scanlines_host(o, direct_wgs, c->io_down));
        HIPCHK(c, hipEventRecord(c->io_ev[6 + s], c->io_down));
        return ARTGPU_OK;
    }
    if (out->on_device) { o.out = static_cast<unsigned char *>(out->scanlines); o.out_stride_bytes = (size_t)out->row_stride_bytes; }
    else {
        float *st;
        if ((rc = pool_get(c, P_IO_OUT0 + s, rowb * ih + 16, &st))) return rc;
        o.out = reinterpret_cast<unsigned char *>(st); o.out_stride_bytes = rowb;         // (free again: the wait for turn i - 2's download above)
    }
    HIPCHK(c, launch_scanlines(o, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->io_host + 8 * i, flags, 8 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (!out->on_device) {
        HIPCHK(c, hipEventRecord(c->io_ev[4 + s], c->stream));
        HIPCHK(c, hipStreamWaitEvent(s_down, c->io_ev[4 + s], 0));
        HIPCHK(c, hipMemcpy2DAsync(out->scanlines, (size_t)out->row_stride_bytes, o.out, rowb, rowb, ih, hipMemcpyDeviceToHost, s_down));
        HIPCHK(c, hipEventRecord(c->io_ev[6 + s], s_down));
    }
    return ARTGPU_OK;
}
} // namespace

int artgpu_batch_run_io(artgpu_ctx *ctx, int nframes, const artgpu_sensor_frame *in, const artgpu_pipeline_params *params, artgpu_scanline_frame *out)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (nframes < 0 || (nframes && (!in || !params || !out))) return fail(ctx, ARTGPU_EINVAL, "batch_run_io: null argument");
    if (nframes == 0) return ARTGPU_OK;
    const int L = std::min(ctx->batch_lanes, nframes);
    int rc = batch_prepare_lanes(ctx, L);          // (L == 1: no lanes, frames_in_flight = 1)
    if (rc) return rc;
    InFlightReset in_flight_reset{ctx};
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));      // inputs the caller produced on this context's stream
    auto px_of = [&](int f) { return (long long)in[f].w * in[f].h; };
    for (int f = 0; f < nframes; ++f) { out[f].status = ARTGPU_OK; out[f].chmax[0] = out[f].chmax[1] = out[f].chmax[2] = out[f].chmax[3] = 0.f; }
    rc = batch_on_lanes(ctx, L, [&](artgpu_ctx *c, int k) -> int {
        const int mine = (nframes - k + L - 1) / L;
        int rc2 = io_setup(c, mine);
        for (int f = k, i = 0; !rc2 && f < nframes; f += L, ++i) {
            rc2 = batch_settle_scratch(c, px_of(f), f + L < nframes ? px_of(f + L) : -1);
            if (!rc2) rc2 = io_frame(c, i, &in[f], params, &out[f]);
        }
        // the lane's copies and kernels, then what the frames left in the pinned words
        c->cu_reserve = 0;
        hipError_t e = hipStreamSynchronize(c->stream);
        if (c->io_down) { const hipError_t e2 = hipStreamSynchronize(c->io_down); if (e == hipSuccess) e = e2; }
        if (c->io_up) { const hipError_t e2 = hipStreamSynchronize(c->io_up); if (e == hipSuccess) e = e2; }
        if (rc2) return rc2;
        if (e != hipSuccess) return fail(c, ARTGPU_EHIP, "batch_run_io: lane %d: %s", k, hipGetErrorString(e));
        if ((rc2 = check_async_faults(c))) return rc2;
        for (int f = k, i = 0; f < nframes; f += L, ++i) {
            const int *w = c->io_host + 8 * i;
            for (int ch = 0; ch < 3; ++ch) std::memcpy(&out[f].chmax[ch], &w[ch], sizeof(float));
            out[f].chmax[3] = out[f].chmax[1];
            if (out[f].rgb2out_enabled && w[4]) {
                out[f].status = ARTGPU_EUNSUPPORTED;
                if (!rc2) rc2 = fail(c, ARTGPU_EUNSUPPORTED, "batch_run_io: frame %d: %d channel values above 1 need ARTOutputProfile::eval (lcms2/libm) on the host", f, w[4]);
            }
        }
        return rc2;
    });
    return rc;
}

// The one collective of a multi-GPU batch: an all-gather of the ranks' completion records over the caller's RCCL communicator.
// RCCL is bound at run time (dlopen), so the library has no link-time dependency on it and single-GPU users never load it.
int artgpu_batch_complete(artgpu_ctx *ctx, void *rccl_comm, int nranks, const int64_t record[ARTGPU_BATCH_RECORD_WORDS], int64_t *all_records)
{
    if (!ctx) return ARTGPU_EINVAL;
    if (!record || !all_records || nranks < 1) return fail(ctx, ARTGPU_EINVAL, "batch_complete: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));      // the record describes work that has finished
    if (!rccl_comm) {
        if (nranks != 1) return fail(ctx, ARTGPU_EINVAL, "batch_complete: %d ranks need a communicator", nranks);
        std::memcpy(all_records, record, sizeof(int64_t) * ARTGPU_BATCH_RECORD_WORDS);
        return ARTGPU_OK;
    }
    // ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t, ncclComm_t, hipStream_t)
    typedef int (*allgather_fn)(const void *, void *, size_t, int, void *, hipStream_t);
    typedef const char *(*errstr_fn)(int);
    static allgather_fn allgather = nullptr;
    static errstr_fn errstr = nullptr;
    if (!allgather) {
        void *h = nullptr;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) return fail(ctx, ARTGPU_EUNSUPPORTED, "batch_complete: cannot load librccl (%s)", dlerror());
        allgather = reinterpret_cast<allgather_fn>(dlsym(h, "ncclAllGather"));
        errstr = reinterpret_cast<errstr_fn>(dlsym(h, "ncclGetErrorString"));
        if (!allgather) return fail(ctx, ARTGPU_EUNSUPPORTED, "batch_complete: librccl has no ncclAllGather");
    }
    constexpr int NCCL_INT64 = 4;       // ncclDataType_t (nccl.h): ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3, ncclInt64 4
    const size_t words = ARTGPU_BATCH_RECORD_WORDS;
    float *buf;
    int rc = pool_get(ctx, P_BATCH, (size_t)(nranks + 1) * words * sizeof(int64_t), &buf);
    if (rc) return rc;
    int64_t *d_send = reinterpret_cast<int64_t *>(buf), *d_recv = d_send + words;
    HIPCHK(ctx, hipMemcpyAsync(d_send, record, words * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    const int nrc = allgather(d_send, d_recv, words, NCCL_INT64, rccl_comm, ctx->stream);
    if (nrc != 0) return fail(ctx, ARTGPU_EHIP, "batch_complete: ncclAllGather failed: %s", errstr ? errstr(nrc) : "?");
    HIPCHK(ctx, hipMemcpyAsync(all_records, d_recv, (size_t)nranks * words * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ARTGPU_OK;
}

} // extern "C"
