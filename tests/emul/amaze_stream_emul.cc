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
        if (rr < 8 || rr >= a.rr1 - 8) continue;
        float h[72];
        for (int j = 0; j < 72; ++j) h[j] = p9_new_weight(lds, a, rr, j);
        for (int j = 0; j < 72; ++j) p9_site(lds, a, rr, j, h[j]);
    }
}
static void wave_p13(float *lds, const TileArgs &a, int r)
{
    for (int rr = r; rr < r + 2; ++rr) {
        if (rr < 10 || rr >= a.rr1 - 10) continue;
        float h[72];
        for (int j = 0; j < 72; ++j) h[j] = p13_new_weight(lds, a, rr, j);
        for (int j = 0; j < 72; ++j) p13_site(lds, a, rr, j, h[j]);
    }
}
static void wave_list(float *lds, const TileArgs &a, int T, int r)
{
    const int buf = (T + 1) & 1;
    int *red = (int *)(lds + RED_OFF);
    unsigned short *list = (unsigned short *)(lds + LIST_OFF + buf * LIST_INTS);
    int n = 0;
    if (r + 1 >= 8 && r < a.rr1 - 8)
        for (int c = 0; c < TS; ++c) {
            int rr;
            if (nyq_site(lds, a, r, c, &rr)) list[n++] = (unsigned short)((rr << 8) | c);
        }
    red[16 + buf] = n;
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

// Streams the tiles (tops[k], lefts[k]), k < ntiles, as ONE workgroup's sequence.  redo_box: nullptr, or four ints per tile -- a tile
// whose box is not all zero... see redo[]: redo[k] != 0 streams tile k with the given TRUE Nyquist box (second attempt).
// valid[k] = 1 if every Nyquist site tile k processed lies inside its true box, box_out[4k..] = that box;
// info[1] = ring tag errors, info[2..4] = first tag error (ring, wanted row, row found)
int amaze_stream_emul_seq(const float *raw, long rs, int W, int H, unsigned filters, float clip_pt, float clip_pt8,
                          int ntiles, const int *tops, const int *lefts, const int *redo, const int *redo_box,
                          float *red, float *green, float *blue, long os, int order, int *valid, int *box_out, long long *info)
{
    TileArgs frame;
    frame.raw = raw; frame.rs = rs; frame.red = red; frame.green = green; frame.blue = blue; frame.os = os;
    frame.top = 0; frame.left = 0; frame.rr1 = 0; frame.gbase = 0; frame.rbase = 0; frame.g0 = 0;
    frame.ny_box = 0;
    frame.W = W; frame.H = H; frame.filters = filters; frame.clip_pt = clip_pt; frame.clip_pt8 = clip_pt8;
    frame.g00 = (int)(fc(filters, 0, 0) & 1);
    if (fc(filters, 0, 0) == 1) frame.ey = fc(filters, 0, 1) == 0 ? 0 : 1;
    else frame.ey = fc(filters, 0, 0) == 0 ? 0 : 1;
    // order & 256: keep the LDS contents of the previous call (what a second kernel launch on the same CU sees) instead of NaN
    static std::vector<float> ldsv(LDS_FLOATS, NAN);
    float *lds = ldsv.data();
    if (!(order & 256)) for (int i = 0; i < LDS_FLOATS; ++i) lds[i] = NAN;
    if (order & 512) {      // arbitrary garbage (what another kernel left in the CU's LDS)
        unsigned sd = 777u + (unsigned)order;
        unsigned *u = (unsigned *)lds;
        for (int i = 0; i < LDS_FLOATS; ++i) { sd = sd * 1664525u + 1013904223u; u[i] = (sd >> 3) & 1 ? sd : (sd & 0x0101ffffu); }
    }
    order &= 255;
    memset(&g_tags, 0, sizeof g_tags);
    for (int k = 0; k < R_COUNT; ++k) for (int sl = 0; sl < 64; ++sl) g_tags.row[k][sl] = -1000;
    std::vector<ThreadRegs> regs(NTHREADS);
    std::vector<P8Regs> p8(64);
    for (auto &x : p8) x.cc = -1;
    for (auto &x : regs) bb_reset(x.bb);
    int p8w = 0;                                   // the wave that plays the P8 role
    for (int w = 0; w < NTHREADS / 64; ++w) if (wave_role(w).a == A_P8) p8w = w;
    std::vector<int> ord(NTHREADS);
    for (int i = 0; i < NTHREADS; ++i) ord[i] = order == 1 ? NTHREADS - 1 - i : i;
    if (order >= 2) {
        unsigned sd = 12345u + (unsigned)order;
        for (int i = NTHREADS - 1; i > 0; --i) { sd = sd * 1664525u + 1013904223u; int j = (int)((sd >> 8) % (unsigned)(i + 1)); int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    }
    auto tile_ref = [&](int k) {
        TileRef t;
        if (k < ntiles && tops[k] > -1000) {           // (top <= -1000: an empty position, as when the redo queue had nothing to offer)
            tile_ref_set(t, k, k, tops[k], lefts[k], (tops[k] + TS < H + 16 ? tops[k] + TS : H + 16) - tops[k]);
            if (redo && redo[k]) { t.tile |= amz::TILE_REDO; t.box = amz::ny_pack(redo_box[4 * k], redo_box[4 * k + 1], redo_box[4 * k + 2], redo_box[4 * k + 3]); }
        } else {
            tile_ref_none(t, k);
        }
        return t;
    };
    TileSeq q;
    tile_ref_none(q.back, -1);
    q.front = tile_ref(0);
    q.next = tile_ref(1);
    seq_begin(lds, 0);
    for (int tid = 0; tid < NTHREADS; ++tid) { const WaveRole wr = wave_role(tid >> 6); if (wr.a == LOADER_ROLE) st_load_first(lds, frame, q, wr.apart * 64 + (tid & 63)); if (wr.a == A_LIGHT) pos_init(lds, wr.apart * 64 + (tid & 63), regs[tid]); }
    const int nsteps = STEPS_PER_TILE * ntiles + TAIL_STEPS;
    for (int T = 0; T < nsteps; ++T) {
        q.t2 = 2 * T;
        if (T > 0 && T % STEPS_PER_TILE == 0) { q.back = q.front; q.front = q.next; q.next = tile_ref(T / STEPS_PER_TILE + 1); }
        if (tile_done(q, T)) {
            const int kb = q.back.gbase / TS, par = kb & 1;
            valid[kb] = tile_valid(lds, par, q.back.rr1, box_out + 4 * kb) ? 1 : 0;
            red_reset(lds, par);
        }
        for (int i = 0; i < NTHREADS; ++i) {
            const WaveRole wr = wave_role(ord[i] >> 6);
            if (wr.a != A_P8) substep_a(lds, frame, q, T, wr.a, wr.apart * 64 + (ord[i] & 63), regs[ord[i]]);
            else p8_step_a(lds, frame, q, T, ord[i] & 63, p8[ord[i] & 63], regs[ord[i]].bb);
        }
        // ---- barrier ----
        const TileArgs a9 = stage_tile(frame, q, 2 * T - 26), al = stage_tile(frame, q, 2 * T - 20);
        const TileArgs ah = stage_tile(frame, q, 2 * T - 30);
        auto workers = [&]() {
            for (int l = 0; l < 64; ++l) p14_worker(lds, ah, ah.rbase - 30, order & 1 ? 63 - l : l);
            for (int l = 0; l < 64; ++l) p10_worker(lds, ah, ah.rbase - 30, order & 1 ? 63 - l : l);
            hot_reset(lds, 0); hot_reset(lds, 1);
        };
        if (order & 1) {
            workers();
            wave_list(lds, al, T, al.rbase - 20);
            for (int l = 63; l >= 0; --l) p8_step_b(lds, frame, q, T, l, p8[l], regs[64 * p8w + l].bb);
            wave_p13(lds, a9, a9.rbase - 26); wave_p9(lds, a9, a9.rbase - 26);
        }
        for (int i = 0; i < NTHREADS; ++i) { const WaveRole wr = wave_role(ord[i] >> 6); if (wr.b < B_P9) substep_b_threads(lds, frame, q, T, wr.b, wr.bpart * 64 + (ord[i] & 63)); }
        if (!(order & 1)) {
            wave_p9(lds, a9, a9.rbase - 26); wave_p13(lds, a9, a9.rbase - 26);
            for (int l = 0; l < 64; ++l) p8_step_b(lds, frame, q, T, l, p8[l], regs[64 * p8w + l].bb);
            wave_list(lds, al, T, al.rbase - 20);
            workers();
        }
        // ---- barrier ----
        if (ntiles == 1) shadow_update(lds);
    }
    info[1] = g_tags.errors; info[2] = g_tags.first_ring; info[3] = g_tags.first_want; info[4] = g_tags.first_have;
    return 0;
}

int amaze_stream_emul_lds_bytes(void) { return LDS_FLOATS * 4; }
}
