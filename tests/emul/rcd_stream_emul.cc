// tests/emul/rcd_stream_emul.cc -- sequential CPU emulation of the RCD row-streaming schedule.
//
// TEST HARNESS ONLY (never linked into libartgpu.so, never used by bench.py's timed region): it compiles the stage code of
// art_amd/csrc/rcd_stream_core.h for the host and executes the workgroup's threads one after the other between two barriers, in a
// caller-chosen order.  Every cross-thread dependency of the schedule has to go through a barrier, so any order must give the
// oracle's bits; the ring-slot tags additionally prove that every consumed read finds the tile row it expects (ring depths, stage
// lags).  LDS starts as NaN / arbitrary garbage and is NOT cleared between tiles.
#define RCS_EMUL 1
#include "../../art_amd/csrc/rcd_stream_core.h"
#include <stdlib.h>
#include <vector>

namespace rcs {
int *g_tag;
long long g_tag_errors, g_tag_first[4];
int g_seq;
}
using namespace rcs;

template <int R>
static void run(const float *raw, long rs, int W, int H, unsigned filters, float *red, float *green, float *blue, long os, int order, long long *info)
{
    typedef Cfg<R> C;
    typedef Sched<R> S;
    std::vector<float> lds_v(C::LDS_FLOATS);
    std::vector<int> tag(C::LDS_FLOATS, 0);
    float *lds = lds_v.data();
    g_tag = tag.data();
    g_tag_errors = 0;
    unsigned seed = 12345u;
    for (auto &x : lds_v) {
        if (order & 512) { seed = seed * 1664525u + 1013904223u; uint32_t b = seed; memcpy(&x, &b, 4); }
        else x = NAN;
    }
    std::vector<int> perm(C::NT);
    for (int i = 0; i < C::NT; ++i) perm[i] = i;
    if ((order & 255) == 1) for (int i = 0; i < C::NT; ++i) perm[i] = C::NT - 1 - i;
    if ((order & 255) == 2) for (int i = C::NT - 1; i > 0; --i) { seed = seed * 1664525u + 1013904223u; int j = (seed >> 8) % (i + 1); std::swap(perm[i], perm[j]); }
    const int numTh = H / TSN + ((H % TSN) ? 1 : 0), numTw = W / TSN + ((W % TSN) ? 1 : 0);
    std::vector<LoadRegs> regs(C::NT);
    long long iters = 0;
    g_seq = 0;
    for (int tr = 0; tr < numTh; ++tr)
        for (int tc = 0; tc < numTw; ++tc) {
            const int rowStart = tr * TSN, rowEnd = rowStart + TS < H ? rowStart + TS : H;
            const int colStart = tc * TSN, colEnd = colStart + TS < W ? colStart + TS : W;
            if (rowStart + BORDER == rowEnd - BORDER || colStart + BORDER == colEnd - BORDER) continue;
            if (rowEnd - rowStart <= 2 * BORDER || colEnd - colStart <= 2 * BORDER) continue;     // writes no pixel
            ++g_seq;
            Tile tl;
            tl.raw = raw + (long)rowStart * rs + colStart; tl.rs = rs;
            tl.red = red + (long)rowStart * os + colStart; tl.green = green + (long)rowStart * os + colStart; tl.blue = blue + (long)rowStart * os + colStart;
            tl.os = os; tl.rows = rowEnd - rowStart; tl.cols = colEnd - colStart; tl.filters = filters; tl.vec2 = 0;
#define ALL(body) for (int n = 0; n < C::NT; ++n) { const int tid = perm[n]; const S s(tid >> 6, tid & 63); body; }
            ALL(s.fetch(tl, 0, regs[tid]); s.commit(lds, tl, 0, regs[tid]))
            for (int A = R; S::more(tl, A); A += R) {
                ++iters;
                ALL(s.fetch(tl, A, regs[tid]); s.i1(lds, tl, A))
                ALL(s.i2(lds, tl, A))
                ALL(s.i3(lds, tl, A))
                ALL(s.i4(lds, tl, A); s.commit(lds, tl, A, regs[tid]))
            }
#undef ALL
        }
    info[0] = iters;
    info[1] = g_tag_errors;
    for (int k = 0; k < 4; ++k) info[2 + k] = g_tag_first[k];
    info[6] = C::LDS_FLOATS * 4;
    info[7] = C::NT;
}

extern "C" int rcd_stream_emul(const float *raw, long rs, int W, int H, unsigned filters, float *red, float *green, float *blue, long os, int R, int order,
                               long long *info)
{
    switch (R) {
    case 4: run<4>(raw, rs, W, H, filters, red, green, blue, os, order, info); return 0;
    case 8: run<8>(raw, rs, W, H, filters, red, green, blue, os, order, info); return 0;
    }
    return -1;
}
