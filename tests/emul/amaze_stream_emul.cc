// tests/emul/amaze_stream_emul.cc -- sequential CPU emulation of the AMaZE v2 streaming schedule.
//
// TEST HARNESS ONLY (never linked into libartgpu.so, never used by bench.py's timed region): it compiles the stage code of
// art_amd/csrc/amaze_stream_core.h for the host and executes the workgroup's 1024 threads one after the other between the two
// barriers of a step, in a caller-chosen order.  Because every cross-thread dependency of the schedule must go through a
// barrier, any order has to give the oracle's bits; the ring-slot tags additionally prove that every consumed read finds the
// tile row it expects (ring depths, stage offsets).  LDS starts as NaN so that a consumed read of a never-written slot shows.
#define AMZ_EMUL 1
#include "../../art_amd/csrc/amaze_stream_core.h"
#include <stdlib.h>
#include <vector>

using namespace amz;

static void wave_p9(float *lds, const TileArgs &a, int r)
{
    for (int rr = r; rr < r + 2; ++rr) {
        if (rr < 8 || rr >= TS - 8) continue;
        float h[72];
        for (int j = 0; j < 72; ++j) h[j] = p9_new_weight(lds, a, rr, j);
        for (int j = 0; j < 72; ++j) p9_site(lds, a, rr, j, h[j]);
    }
}
static void wave_p13(float *lds, const TileArgs &a, int r)
{
    for (int rr = r; rr < r + 2; ++rr) {
        if (rr < 10 || rr >= TS - 10) continue;
        float h[72];
        for (int j = 0; j < 72; ++j) h[j] = p13_new_weight(lds, a, rr, j);
        for (int j = 0; j < 72; ++j) p13_site(lds, a, rr, j, h[j]);
    }
}
static void wave_list(float *lds, const TileArgs &a, int t)
{
    const int r = 2 * t - 20, buf = (t + 1) & 1;
    int *red = (int *)(lds + RED_OFF), *list = (int *)(lds + LIST_OFF + buf * LIST_INTS);
    int n = 0;
    if (r + 1 >= 8 && r < TS - 8)
        for (int c = 0; c < TS; ++c) {
            int rr;
            if (nyq_site(lds, a, r, c, &rr)) list[n++] = (rr << 8) | c;
        }
    red[8 + buf] = n;
}

static float g_shadow[R_COUNT][TS][TS];
static const int ring_off[R_COUNT] = {
#define X(n, d, s) n##_OFF,
    AMZ_RINGS(X)
#undef X
};
static const int ring_depth[R_COUNT] = {
#define X(n, d, s) d,
    AMZ_RINGS(X)
#undef X
};
static const int ring_stride[R_COUNT] = {
#define X(n, d, s) s,
    AMZ_RINGS(X)
#undef X
};
static void shadow_update(const float *lds)
{
    for (int k = 0; k < R_COUNT; ++k)
        for (int sl = 0; sl < ring_depth[k]; ++sl) {
            const int row = g_tags.row[k][sl];
            if (row >= 0 && row < TS) memcpy(g_shadow[k][row], lds + ring_off[k] + sl * ring_stride[k], ring_stride[k] * sizeof(float));
        }
}

extern "C" {
const float *amaze_stream_emul_shadow(int ring) { return &g_shadow[ring][0][0]; }

// info[0] = 1 if the tile is valid for the stream (else the arena kernel has to redo it), info[1] = ring tag errors,
// info[2..4] = first tag error (ring, wanted row, row found)
int amaze_stream_emul_tile(const float *raw, long rs, int W, int H, unsigned filters, float clip_pt, float clip_pt8,
                           int top, int left, float *red, float *green, float *blue, long os, int order, long long *info)
{
    TileArgs a;
    a.raw = raw; a.rs = rs; a.red = red; a.green = green; a.blue = blue; a.os = os;
    a.top = top; a.left = left; a.W = W; a.H = H; a.filters = filters; a.clip_pt = clip_pt; a.clip_pt8 = clip_pt8;
    a.g00 = (int)(fc(filters, 0, 0) & 1);
    if (fc(filters, 0, 0) == 1) a.ey = fc(filters, 0, 1) == 0 ? 0 : 1;
    else a.ey = fc(filters, 0, 0) == 0 ? 0 : 1;
    std::vector<float> ldsv(LDS_FLOATS);
    float *lds = ldsv.data();
    for (int i = 0; i < LDS_FLOATS; ++i) lds[i] = NAN;
    memset(&g_tags, 0, sizeof g_tags);
    for (int k = 0; k < R_COUNT; ++k) for (int s = 0; s < 64; ++s) g_tags.row[k][s] = -1000;
    std::vector<ThreadRegs> regs(NTHREADS);
    std::vector<P8Regs> p8(64);
    for (auto &q : p8) { q.cc = -1; bb_reset(q.bb); }
    for (auto &q : regs) bb_reset(q.bb);
    std::vector<int> ord(NTHREADS);
    for (int i = 0; i < NTHREADS; ++i) ord[i] = order == 1 ? NTHREADS - 1 - i : i;
    if (order >= 2) {
        unsigned s = 12345u + (unsigned)order;
        for (int i = NTHREADS - 1; i > 0; --i) { s = s * 1664525u + 1013904223u; int j = (int)((s >> 8) % (unsigned)(i + 1)); int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    }
    tile_begin(lds, 0);
    for (int tid = 192; tid < 192 + 192; ++tid) st_load_first(a, tid - 192, regs[tid]);
    for (int t = 0; t < NSTEPS; ++t) {
        for (int i = 0; i < NTHREADS; ++i) {
            if (ord[i] < 960) substep_a(lds, a, t, ord[i] / 192, ord[i] % 192, regs[ord[i]]);
            else p8_wave_a(lds, t, ord[i] - 960, p8[ord[i] - 960]);
        }
        // ---- barrier ----
        if (order & 1) {
            wave_list(lds, a, t);
            for (int l = 63; l >= 0; --l) p8_wave_b(lds, t, l, p8[l]);
            wave_p13(lds, a, 2 * t - 26); wave_p9(lds, a, 2 * t - 26);
            for (int i = 0; i < 192; ++i) st_p7(lds, a, 2 * t - 14, i);
        }
        for (int i = 0; i < NTHREADS; ++i) if (ord[i] < 768) substep_b_threads(lds, a, t, ord[i] / 192, ord[i] % 192);
        if (!(order & 1)) {
            for (int i = 0; i < 192; ++i) st_p7(lds, a, 2 * t - 14, i);
            wave_p9(lds, a, 2 * t - 26); wave_p13(lds, a, 2 * t - 26);
            for (int l = 0; l < 64; ++l) p8_wave_b(lds, t, l, p8[l]);
            wave_list(lds, a, t);
        }
        // ---- barrier ----
        shadow_update(lds);
    }
    for (int tid = 192; tid < 384; ++tid) bb_flush(lds, 0, regs[tid].bb);
    for (int l = 0; l < 64; ++l) bb_flush(lds, 4, p8[l].bb);
    info[0] = tile_valid(lds) ? 1 : 0;
    info[1] = g_tags.errors; info[2] = g_tags.first_ring; info[3] = g_tags.first_want; info[4] = g_tags.first_have;
    return 0;
}

int amaze_stream_emul_lds_bytes(void) { return LDS_FLOATS * 4; }
}
