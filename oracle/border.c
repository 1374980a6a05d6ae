/*
 * oracle/border.c -- CPU restatement of RawImageSource::border_interpolate2
 * (reference: rtengine/demosaic_algos.cc:200-353).
 *
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (no reference tests; TU not buildable here).
 *
 * For every pixel of the `bord`-wide frame: 3x3 neighbourhood clipped to the image,
 * per-colour running sums in raster order (i1 outer, j1 inner -- the fp32 summation
 * order is part of the result), native channel copied, the other two = sum/count.
 * The reference visits left columns, right columns (all rows), then top rows and
 * bottom rows restricted to [bord, width-bord); the regions are disjoint except when
 * width < 2*bord, where later writes win -- the same order is kept here.
 */
#include "oracle.h"
#include "oracle_common.h"

static void border_px(int i, int j, int W, int H, const float *raw, size_t rs, unsigned filters,
                      float *red, float *green, float *blue, size_t os)
{
    float sum[6] = {0, 0, 0, 0, 0, 0};
    for (int i1 = i - 1; i1 < i + 2; i1++)
        for (int j1 = j - 1; j1 < j + 2; j1++)
            if (i1 > -1 && i1 < H && j1 > -1 && j1 < W) {
                int c = fc(filters, i1, j1);
                sum[c] += raw[(size_t)i1 * rs + j1];
                sum[c + 3]++;
            }
    int c = fc(filters, i, j);
    size_t o = (size_t)i * os + j;
    float v = raw[(size_t)i * rs + j];
    if (c == 1) {
        red[o] = sum[0] / sum[3];
        green[o] = v;
        blue[o] = sum[2] / sum[5];
    } else {
        green[o] = sum[1] / sum[4];
        if (c == 0) {
            red[o] = v;
            blue[o] = sum[2] / sum[5];
        } else {
            red[o] = sum[0] / sum[3];
            blue[o] = v;
        }
    }
}

void oracle_border_interpolate2(int W, int H, int bord, const float *raw, size_t rs, unsigned filters,
                                float *red, float *green, float *blue, size_t os)
{
    for (int i = 0; i < H; i++) {
        for (int j = 0; j < bord && j < W; j++) border_px(i, j, W, H, raw, rs, filters, red, green, blue, os);
        for (int j = (W - bord > 0 ? W - bord : 0); j < W; j++) border_px(i, j, W, H, raw, rs, filters, red, green, blue, os);
    }
    for (int i = 0; i < bord && i < H; i++)
        for (int j = bord; j < W - bord; j++) border_px(i, j, W, H, raw, rs, filters, red, green, blue, os);
    for (int i = (H - bord > 0 ? H - bord : 0); i < H; i++)
        for (int j = bord; j < W - bord; j++) border_px(i, j, W, H, raw, rs, filters, red, green, blue, os);
}
