/*
 * oracle/xtrans.c -- CPU oracle for the X-Trans (Markesteijn) demosaic, 1-pass/YPbPr and 3-pass/CIELab.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * Restates RawImageSource::xtrans_interpolate (rtengine/xtrans_demosaic.cc:181-969), cielab (L41-116, x86-64 path:
 * 4-lane groups round with cvtps2dq, the scalar tail truncates 0.5 + sum) and xtransborder_interpolate (L122-173).
 * The per-tile work buffer keeps the reference's layout and ALIASING (L301-308): rgb[ndir][114][114][3] | lab[3] |
 * drv[ndir]; greenminmax and the uint8 homogeneity maps live in the lab planes, the 5x5 sums in the drv planes and the
 * per-pixel maximum in homo[ndir-1].  Reads of never-written homogeneity bytes (3-pass, image rows/cols 8..10) therefore
 * see the bytes of this tile's last lab plane exactly as in the reference.  The reference mallocs the buffer once per
 * thread and never clears it; here every tile starts from an all-zero buffer (differences are confined to partial
 * edge tiles).
 *
 * PARITY UNPINNED: xtrans_demosaic.cc needs rtengine.h/rawimagesource.h (glibmm) to compile.
 */
#include "oracle.h"
#include "oracle_common.h"
#include <stdlib.h>
#include <float.h>
#include <xmmintrin.h>
#include <emmintrin.h>

#define TS 114
#define TSH (TS / 2)

static float *g_cbrt;   /* cielab's static LUT (L43-57): 0x14000 entries */
static void cbrt_init(void)
{
    if (g_cbrt) return;
    float *t = (float *)malloc(sizeof(float) * 0x14000);
    const double eps = 216.0 / 24389.0, kappa = 24389.0 / 27.0;
    for (int i = 0; i < 0x14000; i++) {
        double r = i / 65535.0;
        t[i] = (float)(r > eps ? cbrt(r) : (kappa * r + 16.0) / 116.0);
    }
    g_cbrt = t;
}
static inline float cbrt_lut(int i) { return g_cbrt[i < 0 ? 0 : (i > 0x14000 - 1 ? 0x14000 - 1 : i)]; }
static inline int cvt_rn(float x) { return _mm_cvt_ss2si(_mm_set_ss(x)); }

typedef struct {
    int xtrans[6][6];
    short allhex[2][3][3][8];
    int sgrow, sgcol;
    int RightShift[3];
    float xyz_cam[3][3];
} xt_setup;

static inline int fcolx(const xt_setup *s, int row, int col) { return s->xtrans[row % 6][col % 6]; }
static inline int isgreenx(const xt_setup *s, int row, int col) { return s->xtrans[row % 3][col % 3] & 1; }

static void setup(xt_setup *s, const int xtrans[36], const float rgb_cam[12], int width)
{
    static const short orth[12] = {1, 0, 0, 1, -1, 0, 0, -1, 1, 0, 0, 1};
    static const short patt[2][16] = {{0, 1, 0, -1, 2, 0, -1, 0, 1, 1, 1, -1, 0, 0, 0, 0}, {0, 1, 0, -2, 1, 0, -2, 0, 1, 1, -2, -2, 1, -1, -1, 1}};
    static const float xyz_rgb[3][3] = {{0.412453, 0.357580, 0.180423}, {0.212671, 0.715160, 0.072169}, {0.019334, 0.119193, 0.950227}};
    static const float d65_white[3] = {0.950456, 1, 1.088754};
    for (int i = 0; i < 36; ++i) s->xtrans[i / 6][i % 6] = xtrans[i];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            s->xyz_cam[i][j] = 0;
            for (int k = 0; k < 3; k++) s->xyz_cam[i][j] += xyz_rgb[i][k] * rgb_cam[k * 4 + j] / d65_white[i];
        }
    s->sgrow = s->sgcol = 0;
    memset(s->allhex, 0, sizeof(s->allhex));
    for (int row = 0; row < 3; row++)
        for (int col = 0; col < 3; col++) {
            const int gint = isgreenx(s, row, col);
            for (int ng = 0, d = 0; d < 10; d += 2) {
                if (isgreenx(s, row + orth[d] + 6, col + orth[d + 2] + 6)) ng = 0; else ng++;
                if (ng == 4) { s->sgrow = row; s->sgcol = col; }
                if (ng == gint + 1)
                    for (int c = 0; c < 8; c++) {
                        const int v = orth[d] * patt[gint][c * 2] + orth[d + 1] * patt[gint][c * 2 + 1];
                        const int h = orth[d + 2] * patt[gint][c * 2] + orth[d + 3] * patt[gint][c * 2 + 1];
                        s->allhex[0][row][col][c ^ (gint * 2 & d)] = (short)(h + v * width);
                        s->allhex[1][row][col][c ^ (gint * 2 & d)] = (short)(h + v * TS);
                    }
            }
        }
    for (int row = 0; row < 3; row++) {
        int greencount = 0;
        for (int col = 0; col < 3; col++) greencount += isgreenx(s, row, col);
        s->RightShift[row] = (greencount == 2);
    }
}

static inline float limf(float v, float lo, float hi) { return rt_maxf(lo, rt_minf(v, hi)); }

/* cielab (L41-116) on one direction buffer: rgb = &rgb[d][4][4], width = TS, labWidth = TS-8 */
static void cielab_tile(const xt_setup *s, const float *rgb, float *l, float *a, float *b, int height)
{
    const int width = TS, labWidth = TS - 8;
    for (int i = 0; i < height; i++) {
        int j = 0;
        for (; j < labWidth - 3; j += 4)
            for (int k = 0; k < 4; ++k) {
                const float *p = rgb + (size_t)(i * width + j + k) * 3;
                const float x0 = p[0] * s->xyz_cam[0][0] + p[1] * s->xyz_cam[0][1] + p[2] * s->xyz_cam[0][2];
                const float x1 = p[0] * s->xyz_cam[1][0] + p[1] * s->xyz_cam[1][1] + p[2] * s->xyz_cam[1][2];
                const float x2 = p[0] * s->xyz_cam[2][0] + p[1] * s->xyz_cam[2][1] + p[2] * s->xyz_cam[2][2];
                const float c0 = cbrt_lut(cvt_rn(x0)), c1 = cbrt_lut(cvt_rn(x1)), c2 = cbrt_lut(cvt_rn(x2));
                l[i * labWidth + j + k] = 116.f * c1 - 16.f;
                a[i * labWidth + j + k] = 500.f * (c0 - c1);
                b[i * labWidth + j + k] = 200.f * (c1 - c2);
            }
        for (; j < labWidth; j++) {
            float xyz[3] = {0.5f, 0.5f, 0.5f};
            for (int c = 0; c < 3; c++) {
                const float val = rgb[(size_t)(i * width + j) * 3 + c];
                xyz[0] += s->xyz_cam[0][c] * val;
                xyz[1] += s->xyz_cam[1][c] * val;
                xyz[2] += s->xyz_cam[2][c] * val;
            }
            xyz[0] = cbrt_lut((int)xyz[0]);
            xyz[1] = cbrt_lut((int)xyz[1]);
            xyz[2] = cbrt_lut((int)xyz[2]);
            l[i * labWidth + j] = 116 * xyz[1] - 16;
            a[i * labWidth + j] = 500 * (xyz[0] - xyz[1]);
            b[i * labWidth + j] = 200 * (xyz[1] - xyz[2]);
        }
    }
}

static void hex_minmax(const float *pix, const short *hex, float *mn, float *mx)
{
    float minval = FLT_MAX, maxval = 0.f;
    for (int c = 0; c < 6; c++) {
        const float val = pix[hex[c]];
        minval = minval < val ? minval : val;
        maxval = maxval > val ? maxval : val;
    }
    *mn = minval; *mx = maxval;
}

static void process_tile(const xt_setup *s, const float *raw, int width, int height, int top, int left, int passes, int use_cielab,
                         float *buffer, float *red, float *green, float *blue)
{
    const int ndir = 4 << (passes > 1);
    const short dir[4] = {1, TS, TS + 1, TS - 1};
    float (*rgb)[TS][TS][3] = (float (*)[TS][TS][3])buffer;
    float (*lab)[TS - 8][TS - 8] = (float (*)[TS - 8][TS - 8])(buffer + TS * TS * (ndir * 3));
    float (*drv)[TS - 10][TS - 10] = (float (*)[TS - 10][TS - 10])(buffer + TS * TS * (ndir * 3 + 3));
    uint8_t (*homo)[TS][TS] = (uint8_t (*)[TS][TS])lab;
    float (*gmm)[TSH][2] = (float (*)[TSH][2])lab;            /* greenminmaxtile: {min,max} */
    uint8_t (*homosum)[TS][TS] = (uint8_t (*)[TS][TS])drv;
    uint8_t (*homosummax)[TS] = (uint8_t (*)[TS])homo[ndir - 1];

    int mrow = top + TS < height - 3 ? top + TS : height - 3;
    int mcol = left + TS < width - 3 ? left + TS : width - 3;

    /* green min/max (L320-408) */
    for (int row = top; row < mrow; row++) {
        int leftstart = left;
        for (; leftstart < mcol; leftstart++)
            if (!isgreenx(s, row, leftstart)) break;
        const int coloffset = (s->RightShift[row % 3] == 1 ? 3 : 1 + (fcolx(s, row, leftstart + 1) & 1));
        if (coloffset == 3) {
            const short *hex = s->allhex[0][row % 3][leftstart % 3];
            for (int col = leftstart; col < mcol; col += coloffset)
                hex_minmax(&raw[(size_t)row * width + col], hex, &gmm[row - top][(col - left) >> 1][0], &gmm[row - top][(col - left) >> 1][1]);
        } else {
            int col = leftstart;
            float mn, mx;
            if (coloffset == 2) {
                hex_minmax(&raw[(size_t)row * width + col], s->allhex[0][row % 3][col % 3], &mn, &mx);
                gmm[row - top][(col - left) >> 1][0] = mn; gmm[row - top][(col - left) >> 1][1] = mx;
                col += 2;
            }
            const short *hex = s->allhex[0][row % 3][col % 3];
            for (; col < mcol - 1; col += 3) {
                hex_minmax(&raw[(size_t)row * width + col], hex, &mn, &mx);
                gmm[row - top][(col - left) >> 1][0] = mn; gmm[row - top][(col - left) >> 1][1] = mx;
                gmm[row - top][(col + 1 - left) >> 1][0] = mn; gmm[row - top][(col + 1 - left) >> 1][1] = mx;
            }
            if (col < mcol) {
                hex_minmax(&raw[(size_t)row * width + col], hex, &mn, &mx);
                gmm[row - top][(col - left) >> 1][0] = mn; gmm[row - top][(col - left) >> 1][1] = mx;
            }
        }
    }

    memset(rgb, 0, TS * TS * 3 * sizeof(float));
    for (int row = top; row < mrow; row++)
        for (int col = left; col < mcol; col++) rgb[0][row - top][col - left][fcolx(s, row, col)] = raw[(size_t)row * width + col];
    for (int c = 0; c < 3; c++) memcpy(rgb[c + 1], rgb[0], sizeof *rgb);

    /* green along the 4 directions (L422-475) */
    for (int row = top; row < mrow; row++) {
        int leftstart = left;
        for (; leftstart < mcol; leftstart++)
            if (!isgreenx(s, row, leftstart)) break;
        int coloffset = (s->RightShift[row % 3] == 1 ? 3 : 1 + (fcolx(s, row, leftstart + 1) & 1));
        const int flip = coloffset == 3 ? 0 : 1;
        for (int col = leftstart; col < mcol;) {
            const float *pix = &raw[(size_t)row * width + col];
            const short *hex = s->allhex[0][row % 3][col % 3];
            float color[4];
            color[0] = 0.6796875f * (pix[hex[1]] + pix[hex[0]]) - 0.1796875f * (pix[2 * hex[1]] + pix[2 * hex[0]]);
            color[1] = 0.87109375f * pix[hex[3]] + pix[hex[2]] * 0.12890625f + 0.359375f * (pix[0] - pix[-hex[2]]);
            for (int c = 0; c < 2; c++)
                color[2 + c] = 0.640625f * pix[hex[4 + c]] + 0.359375f * pix[-2 * hex[4 + c]] + 0.12890625f * (2.f * pix[0] - pix[3 * hex[4 + c]] - pix[-3 * hex[4 + c]]);
            for (int c = 0; c < 4; c++)
                rgb[c ^ flip][row - top][col - left][1] = limf(color[c], gmm[row - top][(col - left) >> 1][0], gmm[row - top][(col - left) >> 1][1]);
            if (flip) { col += coloffset; coloffset ^= 3; } else col += 3;
        }
    }

    for (int pass = 0; pass < passes; pass++) {
        if (pass == 1) {
            rgb += 4;
            memcpy(rgb, buffer, 4 * sizeof *rgb);
        }
        /* recalculate green from interpolated values of closer pixels (L483-524) */
        if (pass) {
            for (int row = top + 2; row < mrow - 2; row++) {
                int leftstart = left + 2;
                for (; leftstart < mcol - 2; leftstart++)
                    if (!isgreenx(s, row, leftstart)) break;
                int coloffset = (s->RightShift[row % 3] == 1 ? 3 : 1 + (fcolx(s, row, leftstart + 1) & 1));
                const int flip = coloffset == 3 ? 0 : 1;
                for (int col = leftstart; col < mcol - 2;) {
                    const int f = fcolx(s, row, col);
                    const short *hex = s->allhex[1][row % 3][col % 3];
                    for (int d = 3; d < 6; d++) {
                        float (*rix)[3] = &rgb[(d - 2) ^ flip][row - top][col - left];
                        const float val = 0.33333333f * (rix[-2 * hex[d]][1] + 2 * (rix[hex[d]][1] - rix[hex[d]][f]) - rix[-2 * hex[d]][f]) + rix[0][f];
                        rix[0][1] = limf(val, gmm[row - top][(col - left) >> 1][0], gmm[row - top][(col - left) >> 1][1]);
                    }
                    if (flip) { col += coloffset; coloffset ^= 3; } else col += 3;
                }
            }
        }
        /* red and blue for solitary green pixels (L527-561) */
        {
            const int sgstartcol = (left - s->sgcol + 4) / 3 * 3 + s->sgcol;
            float color[3][6];
            for (int row = (top - s->sgrow + 4) / 3 * 3 + s->sgrow; row < mrow - 2; row += 3)
                for (int col = sgstartcol, h = fcolx(s, row, col + 1); col < mcol - 2; col += 3, h ^= 2) {
                    float (*rix)[3] = &rgb[0][row - top][col - left];
                    float diff[6] = {0.f};
                    for (int i = 1, d = 0; d < 6; d++, i ^= TS ^ 1, h ^= 2) {
                        for (int c = 0; c < 2; c++, h ^= 2) {
                            const float g = rix[0][1] + rix[0][1] - rix[i << c][1] - rix[-i << c][1];
                            color[h][d] = g + rix[i << c][h] + rix[-i << c][h];
                            if (d > 1) diff[d] += sqrf(rix[i << c][1] - rix[-i << c][1] - rix[i << c][h] + rix[-i << c][h]) + sqrf(g);
                        }
                        if (d > 2 && (d & 1))
                            if (diff[d - 1] < diff[d])
                                for (int c = 0; c < 2; c++) color[c * 2][d] = color[c * 2][d - 1];
                        if ((d & 1) || d < 2) {
                            for (int c = 0; c < 2; c++) rix[0][c * 2] = 0.5f * color[c * 2][d];
                            rix += TS * TS;
                        }
                    }
                }
        }
        /* red for blue pixels and vice versa (L564-606) */
        for (int row = top + 3; row < mrow - 3; row++) {
            int leftstart = left + 3;
            for (; leftstart < mcol - 1; leftstart++)
                if (!isgreenx(s, row, leftstart)) break;
            int coloffset = (s->RightShift[row % 3] == 1 ? 3 : 1);
            const int c = ((row - s->sgrow) % 3) ? TS : 1;
            const int h = 3 * (c ^ TS ^ 1);
            const int pairs = coloffset != 3;
            if (pairs) coloffset = fcolx(s, row, leftstart + 1) == 1 ? 2 : 1;
            for (int col = leftstart; col < mcol - 3;) {
                const int f = 2 - fcolx(s, row, col);
                float (*rix)[3] = &rgb[0][row - top][col - left];
                for (int d = 0; d < 4; d++, rix += TS * TS) {
                    const int i = d > 1 || ((d ^ c) & 1) ||
                                  ((fabsf(rix[0][1] - rix[c][1]) + fabsf(rix[0][1] - rix[-c][1])) < 2.f * (fabsf(rix[0][1] - rix[h][1]) + fabsf(rix[0][1] - rix[-h][1]))) ? c : h;
                    rix[0][f] = rix[0][1] + 0.5f * (rix[i][f] + rix[-i][f] - rix[i][1] - rix[-i][1]);
                }
                if (pairs) { col += coloffset; coloffset ^= 3; } else col += 3;
            }
        }
        /* red and blue for 2x2 blocks of green (L609-650) */
        {
            int topstart = top + 2;
            for (; topstart < mrow - 2; topstart++)
                if ((topstart - s->sgrow) % 3) break;
            int leftstart = left + 2;
            for (; leftstart < mcol - 2; leftstart++)
                if ((leftstart - s->sgcol) % 3) break;
            const int coloffsetstart = 2 - (fcolx(s, topstart, leftstart + 1) & 1);
            for (int row = topstart; row < mrow - 2; row++)
                if ((row - s->sgrow) % 3)
                    for (int col = leftstart, coloffset = coloffsetstart; col < mcol - 2; col += coloffset, coloffset ^= 3) {
                        float (*rix)[3] = &rgb[0][row - top][col - left];
                        const short *hex = s->allhex[1][row % 3][col % 3];
                        for (int d = 0; d < ndir; d += 2, rix += TS * TS) {
                            if (hex[d] + hex[d + 1]) {
                                const float g = 3 * rix[0][1] - 2 * rix[hex[d]][1] - rix[hex[d + 1]][1];
                                for (int cc = 0; cc < 4; cc += 2) rix[0][cc] = (g + 2 * rix[hex[d]][cc] + rix[hex[d + 1]][cc]) * 0.33333333f;
                            } else {
                                const float g = 2 * rix[0][1] - rix[hex[d]][1] - rix[hex[d + 1]][1];
                                for (int cc = 0; cc < 4; cc += 2) rix[0][cc] = (g + rix[hex[d]][cc] + rix[hex[d + 1]][cc]) * 0.5f;
                            }
                        }
                    }
        }
    }

    rgb = (float (*)[TS][TS][3])buffer;
    mrow -= top;
    mcol -= left;

    /* derivatives (L657-741) */
    for (int d = 0; d < ndir; d++) {
        if (use_cielab) {
            cielab_tile(s, &rgb[d][4][4][0], &lab[0][0][0], &lab[1][0][0], &lab[2][0][0], mrow - 8);
        } else {
            for (int row = 4; row < mrow - 4; row++)
                for (int col = 4; col < mcol - 4; col++) {
                    const float y = 0.2627f * rgb[d][row][col][0] + 0.6780f * rgb[d][row][col][1] + 0.0593f * rgb[d][row][col][2];
                    lab[0][row - 4][col - 4] = y;
                    lab[1][row - 4][col - 4] = (rgb[d][row][col][2] - y) * 0.56433f;
                    lab[2][row - 4][col - 4] = (rgb[d][row][col][0] - y) * 0.67815f;
                }
        }
        int f = dir[d & 3];
        f = f == 1 ? 1 : f - 8;
        for (int row = 5; row < mrow - 5; row++)
            for (int col = 5; col < mcol - 5; col++) {
                const float *l = &lab[0][row - 4][col - 4], *a = &lab[1][row - 4][col - 4], *b = &lab[2][row - 4][col - 4];
                if (use_cielab) {
                    const float g = 2 * l[0] - l[f] - l[-f];
                    drv[d][row - 5][col - 5] = sqrf(g) + sqrf((2 * a[0] - a[f] - a[-f] + g * 2.1551724f)) + sqrf((2 * b[0] - b[f] - b[-f] - g * 0.86206896f));
                } else {
                    drv[d][row - 5][col - 5] = sqrf(2 * l[0] - l[f] - l[-f]) + sqrf(2 * a[0] - a[f] - a[-f]) + sqrf(2 * b[0] - b[f] - b[-f]);
                }
            }
    }

    /* homogeneity maps (L744-811): the vector and scalar forms agree (min of non-negative values, exact counts) */
    for (int row = 6; row < mrow - 6; row++)
        for (int col = 6; col < mcol - 6; col++) {
            float tr = drv[0][row - 5][col - 5] < drv[1][row - 5][col - 5] ? drv[0][row - 5][col - 5] : drv[1][row - 5][col - 5];
            for (int d = 2; d < ndir; d++) tr = (drv[d][row - 5][col - 5] < tr ? drv[d][row - 5][col - 5] : tr);
            tr *= 8;
            for (int d = 0; d < ndir; d++) {
                uint8_t temp = 0;
                for (int v = -1; v <= 1; v++)
                    for (int h = -1; h <= 1; h++) temp += (drv[d][row + v - 5][col + h - 5] <= tr ? 1 : 0);
                homo[d][row][col] = temp;
            }
        }

    if (height - top < TS + 4) mrow = height - top + 2;
    if (width - left < TS + 4) mcol = width - left + 2;

    /* 5x5 sums (L823-866); the 16-wide vector loop and the running-sum tail give the same sums (no uint8 saturation: 25*9) */
    const int startrow = top < 8 ? top : 8, startcol = left < 8 ? left : 8;
    for (int d = 0; d < ndir; d++)
        for (int row = startrow; row < mrow - 8; row++) {
            const int endcol = row < mrow - 9 ? mcol - 8 : mcol - 23;
            int col = startcol;
            for (; col < endcol; col += 16)            /* writes up to 15 columns past endcol, like the reference */
                for (int k = 0; k < 16; ++k) {
                    int sum = 0;
                    for (int v = -2; v <= 2; v++)
                        for (int h = -2; h <= 2; h++) sum += homo[d][row + v][col + k + h];
                    homosum[d][row][col + k] = (uint8_t)(sum > 255 ? 255 : sum);
                }
            for (; col < mcol - 8; col++) {
                int sum = 0;
                for (int v = -2; v <= 2; v++)
                    for (int h = -2; h <= 2; h++) sum += homo[d][row + v][col + h];
                homosum[d][row][col] = (uint8_t)sum;
            }
        }

    /* per-pixel maximum (L870-906) */
    for (int row = startrow; row < mrow - 8; row++) {
        const int endcol = row < mrow - 9 ? mcol - 8 : mcol - 23;
        int col = startcol;
        for (; col < endcol; col += 16)
            for (int k = 0; k < 16; ++k) {
                uint8_t maxval = homosum[0][row][col + k];
                for (int d = 1; d < ndir; d++) maxval = maxval < homosum[d][row][col + k] ? homosum[d][row][col + k] : maxval;
                maxval -= maxval >> 3;
                homosummax[row][col + k] = maxval;
            }
        for (; col < mcol - 8; col++) {
            uint8_t maxval = homosum[0][row][col];
            for (int d = 1; d < ndir; d++) maxval = maxval < homosum[d][row][col] ? homosum[d][row][col] : maxval;
            maxval -= maxval >> 3;
            homosummax[row][col] = maxval;
        }
    }

    /* average the most homogeneous directions (L910-949) */
    for (int row = startrow; row < mrow - 8; row++)
        for (int col = startcol; col < mcol - 8; col++) {
            uint8_t hm[8] = {0};
            for (int d = 0; d < 4; d++) hm[d] = homosum[d][row][col];
            for (int d = 4; d < ndir; d++) {
                hm[d] = homosum[d][row][col];
                if (hm[d - 4] < hm[d]) hm[d - 4] = 0;
                else if (hm[d - 4] > hm[d]) hm[d] = 0;
            }
            float avg[4] = {0.f};
            const uint8_t maxval = homosummax[row][col];
            for (int d = 0; d < ndir; d++)
                if (hm[d] >= maxval) {
                    for (int c = 0; c < 3; c++) avg[c] += rgb[d][row][col][c];
                    avg[3]++;
                }
            const size_t o = (size_t)(row + top) * width + col + left;
            red[o] = std_maxf(0.f, avg[0] / avg[3]);
            green[o] = std_maxf(0.f, avg[1] / avg[3]);
            blue[o] = std_maxf(0.f, avg[2] / avg[3]);
        }
}

/* xtransborder_interpolate (L122-173) */
void oracle_xtrans_border(const float *raw, int width, int height, const int xtrans[36], int border, float *red, float *green, float *blue)
{
    static const float weight[3][3] = {{0.25f, 0.5f, 0.25f}, {0.5f, 0.f, 0.5f}, {0.25f, 0.5f, 0.25f}};
    for (int row = 0; row < height; row++)
        for (int col = 0; col < width; col++) {
            if (col == border && row >= border && row < height - border) col = width - border;
            float sum[6] = {0.f};
            for (int y = row - 1 > 0 ? row - 1 : 0, v = row == 0 ? 0 : -1; y <= (row + 1 < height - 1 ? row + 1 : height - 1); y++, v++)
                for (int x = col - 1 > 0 ? col - 1 : 0, h = col == 0 ? 0 : -1; x <= (col + 1 < width - 1 ? col + 1 : width - 1); x++, h++) {
                    const int f = xtrans[(y % 6) * 6 + x % 6];
                    sum[f] += raw[(size_t)y * width + x] * weight[v + 1][h + 1];
                    sum[f + 3] += weight[v + 1][h + 1];
                }
            const size_t o = (size_t)row * width + col;
            switch (xtrans[(row % 6) * 6 + col % 6]) {
            case 0:
                red[o] = raw[o]; green[o] = sum[1] / sum[4]; blue[o] = sum[2] / sum[5];
                break;
            case 1:
                if (sum[3] == 0.f) red[o] = green[o] = blue[o] = raw[o];
                else { red[o] = sum[0] / sum[3]; green[o] = raw[o]; blue[o] = sum[2] / sum[5]; }
                break;
            case 2:
                red[o] = sum[0] / sum[3]; green[o] = sum[1] / sum[4]; blue[o] = raw[o];
            }
        }
}

/* test hook: 1 = what ONE reference thread does -- the tile buffer is allocated once and never cleared (xtrans_demosaic.cc:295-315),
   tiles in raster order -- instead of this restatement's "every tile starts from a zeroed buffer" */
int oracle_xtrans_stale = 0;

/* rgb_cam: 3x4 row-major (RawImage::getRgbCam).  Output planes are W x H, fully written (tiles + border). */
void oracle_xtrans_demosaic(const float *raw, int width, int height, const int xtrans[36], const float rgb_cam[12], int passes, int use_cielab,
                            float *red, float *green, float *blue)
{
    xt_setup s;
    setup(&s, xtrans, rgb_cam, width);
    cbrt_init();
    const int ndir = 4 << (passes > 1);
    const size_t nbuf = (size_t)TS * TS * (ndir * 4 + 3) + 128;
    const int ntx = (width - 19 - 3 + (TS - 16) - 1) / (TS - 16), nty = (height - 19 - 3 + (TS - 16) - 1) / (TS - 16);
#pragma omp parallel if (!oracle_xtrans_stale)
    {
        float *buffer = (float *)calloc(nbuf, sizeof(float));
#pragma omp for collapse(2) schedule(dynamic, 2)
        for (int ty = 0; ty < nty; ++ty)
            for (int tx = 0; tx < ntx; ++tx) {
                if (!oracle_xtrans_stale) memset(buffer, 0, nbuf * sizeof(float));
                process_tile(&s, raw, width, height, 3 + ty * (TS - 16), 3 + tx * (TS - 16), passes, use_cielab, buffer, red, green, blue);
            }
        free(buffer);
    }
    oracle_xtrans_border(raw, width, height, xtrans, passes > 1 ? 8 : 11, red, green, blue);
}
