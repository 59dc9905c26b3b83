/*
 * bbme_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see bbme_oracle.h).
 *
 * Plain-C restatement of the reference's hot path.  Every function cites the
 * reference lines it follows (paths relative to the reference repo root).
 * Arithmetic types follow the reference: motion vectors and energies are
 * float32, SADs are int, the padding search runs in double.
 *
 * Hot path (search / regulariser / driver): PARITY UNPINNED -- the reference
 * holds no golden vectors for it and cannot be built here (needs OpenCV).
 * Build with -ffp-contract=off so float expressions round as written.
 */
#include "bbme_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* cv::norm(a, b, NORM_L1) on two 8-bit ROIs, as used at                      */
/* motion_framework.cpp:315,338,354,370,386,404,599: exact integer sum.       */
/* ------------------------------------------------------------------------ */
static int l1_norm_u8(const uint8_t *a, int pitch_a, const uint8_t *b, int pitch_b, int bs)
{
    int acc = 0;
    for (int y = 0; y < bs; ++y) {
        const uint8_t *ra = a + (size_t)y * pitch_a;
        const uint8_t *rb = b + (size_t)y * pitch_b;
        int row = 0;
        for (int x = 0; x < bs; ++x) {
            int d = (int)ra[x] - (int)rb[x];
            row += d < 0 ? -d : d;
        }
        acc += row;
    }
    return acc;
}

/* ------------------------------------------------------------------------ */
/* Padding search: motion_framework.cpp:14-54                                 */
/* ------------------------------------------------------------------------ */
int orc_plan_padding(int width, int height, const int *block_size, int num_levels,
                     int *padded_width, int *padded_height, int *pad_x, int *pad_y)
{
    double temp_h = (double)height;                       /* :15 */
    double temp_w = (double)width;                        /* :16 */
    for (;;) {                                            /* :19 */
        if (temp_h == 2 * height || temp_w == 2 * width)  /* :21 -> message + exit(1) */
            return -1;
        double rem_h = 0, rem_w = 0;
        for (int i = 0; i < num_levels; ++i) {            /* :31-35 */
            rem_h += fmod(temp_h, pow(2, i) * block_size[i]);
            rem_w += fmod(temp_w, pow(2, i) * block_size[i]);
        }
        if (rem_h == 0 && rem_w == 0)                     /* :37 */
            break;
        if (rem_h != 0) temp_h++;                         /* :41-44 */
        if (rem_w != 0) temp_w++;
    }
    *padded_height = (int)temp_h;                         /* :48-49 */
    *padded_width = (int)temp_w;
    *pad_x = ((int)temp_w - width) / 2;                   /* :50-51 */
    *pad_y = ((int)temp_h - height) / 2;
    /* :57-58 allocates (rows+2*pad_y) x (cols+2*pad_x); with an odd difference that
     * is one short of padded_* and no longer block-divisible (the reference then
     * reads out of bounds).  The oracle refuses instead of reproducing UB. */
    if (((int)temp_w - width) % 2 != 0 || ((int)temp_h - height) % 2 != 0)
        return -2;
    return 0;
}

/* cv::copyMakeBorder(..., BORDER_CONSTANT, Scalar(0)) at motion_framework.cpp:60-61 */
void orc_pad_zero(const uint8_t *src, int width, int height, int pitch,
                  int pad_x, int pad_y, uint8_t *dst)
{
    int pw = width + 2 * pad_x, ph = height + 2 * pad_y;
    memset(dst, 0, (size_t)pw * ph);
    for (int y = 0; y < height; ++y)
        memcpy(dst + (size_t)(y + pad_y) * pw + pad_x, src + (size_t)y * pitch, (size_t)width);
}

/* cv::pyrDown(src, dst, Size(cols/2, rows/2)) at motion_framework.cpp:89-90.
 * OpenCV 8-bit path (imgproc/pyramids.cpp): separable [1 4 6 4 1], integer, no
 * rounding between passes, (sum + 128) >> 8, BORDER_REFLECT_101, dst(x,y) centred
 * on src(2x,2y).  Restated from OpenCV's algorithm -- PARITY UNPINNED. */
static int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

void orc_pyr_down(const uint8_t *src, int sw, int sh, uint8_t *dst)
{
    static const int w5[5] = {1, 4, 6, 4, 1};
    int dw = sw / 2, dh = sh / 2;
    int *rows = (int *)malloc(sizeof(int) * (size_t)dw * 5);
    for (int y = 0; y < dh; ++y) {
        for (int k = 0; k < 5; ++k) {
            int sy = reflect101(2 * y + k - 2, sh);
            const uint8_t *srow = src + (size_t)sy * sw;
            int *r = rows + (size_t)k * dw;
            for (int x = 0; x < dw; ++x) {
                int s = 0;
                for (int t = 0; t < 5; ++t)
                    s += w5[t] * srow[reflect101(2 * x + t - 2, sw)];
                r[x] = s;
            }
        }
        for (int x = 0; x < dw; ++x) {
            int s = 0;
            for (int k = 0; k < 5; ++k)
                s += w5[k] * rows[(size_t)k * dw + x];
            dst[(size_t)y * dw + x] = (uint8_t)((s + 128) >> 8);
        }
    }
    free(rows);
}

/* cv::resize(img, img, Size(), 4, 4, INTER_LINEAR) at main_class.cpp:32-33.
 * OpenCV 8-bit fixed-point bilinear (imgproc/imgwarp.cpp: 11-bit coefficients,
 * HResizeLinear to int, VResizeLinear<uchar> with the >>4, >>16, +2, >>2 chain).
 * Restated from OpenCV's algorithm -- PARITY UNPINNED; outside the hot path. */
static short sat_short_round(float v)
{
    long r = lrintf(v);
    if (r > 32767) r = 32767;
    if (r < -32768) r = -32768;
    return (short)r;
}

void orc_resize_linear_x4(const uint8_t *src, int sw, int sh, uint8_t *dst)
{
    const int scale = 4;
    int dw = sw * scale, dh = sh * scale;
    double inv = 1.0 / scale;
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dw);
    short *alpha = (short *)malloc(sizeof(short) * 2 * (size_t)dw);
    int *yofs = (int *)malloc(sizeof(int) * (size_t)dh);
    short *beta = (short *)malloc(sizeof(short) * 2 * (size_t)dh);
    int xmax = dw;
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * inv - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) { if (xmax > dx) xmax = dx; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        alpha[2 * dx] = sat_short_round((1.f - fx) * 2048.f);
        alpha[2 * dx + 1] = sat_short_round(fx * 2048.f);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * inv - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        yofs[dy] = sy;
        beta[2 * dy] = sat_short_round((1.f - fy) * 2048.f);
        beta[2 * dy + 1] = sat_short_round(fy * 2048.f);
    }
    int *r0 = (int *)malloc(sizeof(int) * (size_t)dw);
    int *r1 = (int *)malloc(sizeof(int) * (size_t)dw);
    for (int dy = 0; dy < dh; ++dy) {
        int sy0 = yofs[dy], sy1 = yofs[dy] + 1;
        if (sy0 < 0) sy0 = 0; if (sy0 > sh - 1) sy0 = sh - 1;   /* row clipping as OpenCV's clip() */
        if (sy1 < 0) sy1 = 0; if (sy1 > sh - 1) sy1 = sh - 1;
        const uint8_t *s0 = src + (size_t)sy0 * sw, *s1 = src + (size_t)sy1 * sw;
        for (int dx = 0; dx < dw; ++dx) {
            int sx = xofs[dx];
            if (dx < xmax) {
                r0[dx] = s0[sx] * alpha[2 * dx] + s0[sx + 1] * alpha[2 * dx + 1];
                r1[dx] = s1[sx] * alpha[2 * dx] + s1[sx + 1] * alpha[2 * dx + 1];
            } else {
                r0[dx] = s0[sx] * 2048;
                r1[dx] = s1[sx] * 2048;
            }
        }
        int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        for (int dx = 0; dx < dw; ++dx)
            dst[(size_t)dy * dw + dx] =
                (uint8_t)((((b0 * (r0[dx] >> 4)) >> 16) + ((b1 * (r1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(r0); free(r1); free(xofs); free(alpha); free(yofs); free(beta);
}

/* ------------------------------------------------------------------------ */
/* MF::MF: motion_framework.cpp:4-111                                         */
/* ------------------------------------------------------------------------ */
static void level_alloc(orc_level *lv, int w, int h, int bs, int ss, int use_cache)
{
    lv->width = w; lv->height = h;
    lv->block_size = bs;                                   /* :71,93 */
    lv->search_size = ss;                                  /* :72,94 */
    lv->lambda = (float)(bs / 2);                          /* :73,95 (integer division first) */
    lv->image1 = (uint8_t *)malloc((size_t)w * h);
    lv->image2 = (uint8_t *)malloc((size_t)w * h);
    lv->flow = (float *)calloc((size_t)w * h * 2, sizeof(float));     /* Mat::zeros :70,92 */
    lv->cache = use_cache ? (int32_t *)calloc((size_t)w * h * 4, sizeof(int32_t)) : NULL; /* :77,97 */
}

int orc_mf_create(const uint8_t *image1, const uint8_t *image2, int width, int height, int pitch,
                  const int *search_size, const int *block_size, int num_levels,
                  int use_cache, orc_mf **out)
{
    if (num_levels <= 0) return -3;                        /* assert :7 */
    int pw, ph, px, py;
    int rc = orc_plan_padding(width, height, block_size, num_levels, &pw, &ph, &px, &py);
    if (rc) return rc;
    orc_mf *mf = (orc_mf *)calloc(1, sizeof(orc_mf));
    mf->num_levels = num_levels;
    mf->lv = (orc_level *)calloc((size_t)num_levels, sizeof(orc_level));
    mf->orig_height = height; mf->orig_width = width;      /* :11-12 */
    mf->padded_height = ph; mf->padded_width = pw;         /* :48-49 */
    mf->padding_x = px; mf->padding_y = py;                /* :53-54 */
    mf->lambda_multiplier = 1;                             /* :64 */
    mf->use_cache = use_cache;
    level_alloc(&mf->lv[0], width + 2 * px, height + 2 * py, block_size[0], search_size[0], use_cache);
    orc_pad_zero(image1, width, height, pitch, px, py, mf->lv[0].image1);   /* :60 */
    orc_pad_zero(image2, width, height, pitch, px, py, mf->lv[0].image2);   /* :61 */
    for (int i = 1; i < num_levels; ++i) {                 /* :86-106 */
        orc_level *p = &mf->lv[i - 1];
        level_alloc(&mf->lv[i], p->width / 2, p->height / 2, block_size[i], search_size[i], use_cache);
        orc_pyr_down(p->image1, p->width, p->height, mf->lv[i].image1);     /* :89 */
        orc_pyr_down(p->image2, p->width, p->height, mf->lv[i].image2);     /* :90 */
    }
    *out = mf;
    return 0;
}

int orc_mf_create_from_planes(const uint8_t *const *img1_lv, const uint8_t *const *img2_lv,
                              const int *widths, const int *heights,
                              const int *search_size, const int *block_size, int num_levels,
                              int use_cache, orc_mf **out)
{
    if (num_levels <= 0) return -3;
    orc_mf *mf = (orc_mf *)calloc(1, sizeof(orc_mf));
    mf->num_levels = num_levels;
    mf->lv = (orc_level *)calloc((size_t)num_levels, sizeof(orc_level));
    mf->padded_height = heights[0]; mf->padded_width = widths[0];
    mf->orig_height = heights[0]; mf->orig_width = widths[0];
    mf->lambda_multiplier = 1;
    mf->use_cache = use_cache;
    for (int i = 0; i < num_levels; ++i) {
        level_alloc(&mf->lv[i], widths[i], heights[i], block_size[i], search_size[i], use_cache);
        memcpy(mf->lv[i].image1, img1_lv[i], (size_t)widths[i] * heights[i]);
        memcpy(mf->lv[i].image2, img2_lv[i], (size_t)widths[i] * heights[i]);
    }
    *out = mf;
    return 0;
}

void orc_mf_destroy(orc_mf *mf)
{
    if (!mf) return;
    for (int i = 0; i < mf->num_levels; ++i) {
        free(mf->lv[i].image1); free(mf->lv[i].image2);
        free(mf->lv[i].flow); free(mf->lv[i].cache);
    }
    free(mf->lv);
    free(mf);
}

#define FLOW_AT(lv, y, x) ((lv)->flow + 2 * ((size_t)(y) * (lv)->width + (x)))
#define CACHE_AT(lv, y, x) ((lv)->cache + 4 * ((size_t)(y) * (lv)->width + (x)))

/* ------------------------------------------------------------------------ */
/* MF::fill_block_MV :803-813, MF::copy_to_all_pixels :815-826,               */
/* MF::copyMVs :828-843, MF::divide_blocks :845-862                           */
/* ------------------------------------------------------------------------ */
static void fill_block_mv(orc_level *lv, int i, int j, int bs, float u, float v)
{
    for (int k = i; k < i + bs; ++k)
        for (int l = j; l < j + bs; ++l) {
            float *f = FLOW_AT(lv, k, l);
            f[0] = u; f[1] = v;
        }
}

void orc_copy_to_all_pixels(orc_mf *mf, int level)
{
    orc_level *lv = &mf->lv[level];
    int bs = lv->block_size;
    for (int i = 0; i < lv->height; i += bs)
        for (int j = 0; j < lv->width; j += bs) {
            const float *f = FLOW_AT(lv, i, j);
            fill_block_mv(lv, i, j, bs, f[0], f[1]);
        }
}

void orc_copy_mvs(orc_mf *mf, int level)
{
    orc_level *cur = &mf->lv[level], *prev = &mf->lv[level + 1];
    int bs = prev->block_size;                                       /* :830 */
    for (int i = 0; i < prev->height; i += bs)
        for (int j = 0; j < prev->width; j += bs) {
            const float *f = FLOW_AT(prev, i, j);
            float u = f[0] * 2.0f, v = f[1] * 2.0f;                  /* .mul(Vec2f(2,2)) :836 */
            fill_block_mv(cur, i << 1, j << 1, bs << 1, u, v);        /* :840 */
        }
}

void orc_divide_blocks(orc_mf *mf, int level)
{
    orc_level *lv = &mf->lv[level];
    int bo = lv->block_size, bn = lv->block_size >> 1;
    for (int i = 0; i < lv->height; i += bo)
        for (int j = 0; j < lv->width; j += bo) {
            const float *f = FLOW_AT(lv, i, j);
            float u = f[0], v = f[1];
            float *a = FLOW_AT(lv, i + bn, j);      a[0] = u; a[1] = v;   /* :857 */
            float *b = FLOW_AT(lv, i, j + bn);      b[0] = u; b[1] = v;   /* :858 */
            float *c = FLOW_AT(lv, i + bn, j + bn); c[0] = u; c[1] = v;   /* :859 */
        }
}

/* ------------------------------------------------------------------------ */
/* MF::find_min_block_spiral :296-422 (literal walk, first strict minimum)    */
/* ------------------------------------------------------------------------ */
typedef struct {
    const orc_level *lv;
    int x1, y1, bs, width, height;
    int sad_min, min_x, min_y;
} spiral_state;

static void spiral_visit(spiral_state *s, int l, int k)
{
    /* :335,351,367,383,401 -- candidates leaving the image are skipped, not clamped */
    if (l < 0 || k < 0 || (l + s->bs) > s->width || (k + s->bs) > s->height)
        return;
    int sad = l1_norm_u8(s->lv->image1 + (size_t)s->y1 * s->width + s->x1, s->width,
                         s->lv->image2 + (size_t)k * s->width + l, s->width, s->bs);
    if (sad < s->sad_min) {                                           /* strict :339 */
        s->sad_min = sad; s->min_x = l; s->min_y = k;
    }
}

void orc_find_min_block_spiral(orc_mf *mf, int level, int image1_ypos, int image1_xpos,
                               int image2_ypos, int image2_xpos, int *pos_x, int *pos_y)
{
    orc_level *lv = &mf->lv[level];
    int shift = lv->search_size - lv->block_size;                     /* :299 */
    int bs = lv->block_size, width = lv->width, height = lv->height;

    if (image2_xpos < 0 || image2_ypos < 0 ||
        (image2_xpos + bs) > width || (image2_ypos + bs) > height) {  /* :304-310 */
        *pos_x = image1_xpos; *pos_y = image1_ypos;                   /* MV becomes zero */
        return;
    }
    spiral_state s;
    s.lv = lv; s.x1 = image1_xpos; s.y1 = image1_ypos; s.bs = bs;
    s.width = width; s.height = height;
    s.min_x = image2_xpos; s.min_y = image2_ypos;                     /* :312-313 */
    s.sad_min = l1_norm_u8(lv->image1 + (size_t)image1_ypos * width + image1_xpos, width,
                           lv->image2 + (size_t)image2_ypos * width + image2_xpos, width, bs); /* :315 */
    int l = s.min_x, k = s.min_y, m, t;                               /* :317-319 */
    for (m = 1; m < shift; m += 2) {                                  /* :326 */
        for (t = 0; t < m; ++t)     { l += 1; spiral_visit(&s, l, k); }   /* right m   :331-345 */
        for (t = 0; t < m; ++t)     { k += 1; spiral_visit(&s, l, k); }   /* down m    :347-361 */
        for (t = 0; t < m + 1; ++t) { l -= 1; spiral_visit(&s, l, k); }   /* left m+1  :363-377 */
        for (t = 0; t < m + 1; ++t) { k -= 1; spiral_visit(&s, l, k); }   /* up m+1    :379-393 */
    }
    for (t = 0; t < (m - 1); ++t)   { l += 1; spiral_visit(&s, l, k); }   /* top row   :397-411 */

    if (lv->cache) {                                                  /* :414 */
        int32_t *c = CACHE_AT(lv, image1_ypos, image1_xpos);
        c[0] = s.min_x; c[1] = s.min_y; c[2] = s.sad_min; c[3] = bs;
    }
    *pos_x = s.min_x; *pos_y = s.min_y;
}

/* ------------------------------------------------------------------------ */
/* MF::find_min_block :246-294 -- the raster full search whose call is commented */
/* out at :235.  Window clamped to the image (:260,262), no special case for a   */
/* prediction outside it; among equal SADs the block closer (L1) to the block's  */
/* own position wins (:276-281), the first one in raster order among those.       */
/* ------------------------------------------------------------------------ */
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

void orc_find_min_block(orc_mf *mf, int level, int image1_ypos, int image1_xpos,
                        int image2_ypos, int image2_xpos, int *pos_x, int *pos_y)
{
    orc_level *lv = &mf->lv[level];
    int start_pos = (lv->search_size - lv->block_size) >> 1;          /* :249 */
    int SAD_min = INT_MAX;                                             /* :250 */
    int min_x = image2_xpos, min_y = image2_ypos;                      /* :251-252 */
    int l1_dist = INT_MAX;                                             /* :257 */
    int bs = lv->block_size, width = lv->width, height = lv->height;
    for (int k = imax(0, image2_ypos - start_pos); k < imin(height - bs + 1, image2_ypos + start_pos + 1); k++)      /* :260 */
        for (int l = imax(0, image2_xpos - start_pos); l < imin(width - bs + 1, image2_xpos + start_pos + 1); l++) { /* :262 */
            int sad = l1_norm_u8(lv->image1 + (size_t)image1_ypos * width + image1_xpos, width,
                                 lv->image2 + (size_t)k * width + l, width, bs);                                      /* :265 */
            int d = abs(image1_xpos - l) + abs(image1_ypos - k);
            if (sad < SAD_min) {                                       /* :269-275 */
                SAD_min = sad; min_x = l; min_y = k; l1_dist = d;
            } else if (sad == SAD_min && d < l1_dist) {                /* :276-281 */
                min_x = l; min_y = k; l1_dist = d;
            }
        }
    if (lv->cache) {                                                   /* :286 */
        int32_t *c = CACHE_AT(lv, image1_ypos, image1_xpos);
        c[0] = min_x; c[1] = min_y; c[2] = SAD_min; c[3] = bs;
    }
    *pos_x = min_x; *pos_y = min_y;
}

int orc_spiral_walk(int shift, int *dx, int *dy, int cap)
{
    int n = 0, l = 0, k = 0, m, t;
#define EMIT() do { if (n < cap) { dx[n] = l; dy[n] = k; } ++n; } while (0)
    EMIT();
    for (m = 1; m < shift; m += 2) {
        for (t = 0; t < m; ++t)     { l += 1; EMIT(); }
        for (t = 0; t < m; ++t)     { k += 1; EMIT(); }
        for (t = 0; t < m + 1; ++t) { l -= 1; EMIT(); }
        for (t = 0; t < m + 1; ++t) { k -= 1; EMIT(); }
    }
    for (t = 0; t < (m - 1); ++t)   { l += 1; EMIT(); }
#undef EMIT
    return n;
}

/* MF::calcLevelBM :226-244 */
void orc_calc_level_bm(orc_mf *mf, int level)
{
    orc_level *lv = &mf->lv[level];
    int bs = lv->block_size;
    /* Blocks are independent in calcLevelBM (each reads and writes only its own origin cell), so the all-cores leg of
     * bench.py's cpu_baseline builds this file with -fopenmp; without it the pragma is ignored and the loop is the
     * reference's single thread.  The regulariser sweeps stay sequential (they are order dependent). */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int i = 0; i < lv->height; i += bs)
        for (int j = 0; j < lv->width; j += bs) {
            float *f = FLOW_AT(lv, i, j);
            int x2 = j + (int)f[0];                                   /* :233 */
            int y2 = i + (int)f[1];                                   /* :234 */
            int rx, ry;
            if (mf->raster_search) orc_find_min_block(mf, level, i, j, y2, x2, &rx, &ry);   /* :235 */
            else orc_find_min_block_spiral(mf, level, i, j, y2, x2, &rx, &ry);              /* :236 */
            f[0] = (float)rx - j;                                     /* :238-239 */
            f[1] = (float)ry - i;
        }
}

/* ------------------------------------------------------------------------ */
/* MF::calculate_smoothness :623-644, MF::min_energy_candidate :646-662,      */
/* MF::find_min_candidate :532-621                                            */
/* ------------------------------------------------------------------------ */
static float calculate_smoothness(int current, const float (*cand)[2], int csize)
{
    float cost = 0;
    float mu = cand[current][0], mv = cand[current][1];
    for (int i = 0; i < csize; ++i)
        cost += fabsf(cand[i][0] - mu) + fabsf(cand[i][1] - mv);      /* :640 */
    return cost;
}

static int min_energy_candidate(const float *energy, int esize)
{
    float min_val = energy[0];
    int min_pos = 0;
    for (int i = 1; i < esize; ++i)
        if (energy[i] < min_val) { min_val = energy[i]; min_pos = i; }    /* strict :655 */
    return min_pos;
}

static int eval_candidates(orc_mf *mf, orc_level *lv, int pos_x1, int pos_y1,
                           const float (*cand)[2], int csize)
{
    float energy[9];
    int bs = lv->block_size;
    float lambda = lv->lambda;
    int height = lv->height, width = lv->width;
    float p1x = (float)pos_x1, p1y = (float)pos_y1;                   /* :566 */

    for (int i = 0; i < csize; ++i) {
        float p2x = p1x + cand[i][0], p2y = p1y + cand[i][1];         /* :576 */
        if ((int)p2x < 0 || (int)p2x > (width - bs) ||
            (int)p2y < 0 || (int)p2y > (height - bs)) {               /* :578 */
            energy[i] = FLT_MAX;                                      /* :580 */
            continue;
        }
        int sad;
        int32_t *c = lv->cache ? CACHE_AT(lv, pos_y1, pos_x1) : NULL;
        if (c && c[0] == (int)p2x && c[1] == (int)p2y && c[3] == bs) {    /* :594-596 */
            sad = c[2];
        } else {
            sad = l1_norm_u8(lv->image1 + (size_t)pos_y1 * width + pos_x1, width,
                             lv->image2 + (size_t)(int)p2y * width + (int)p2x, width, bs); /* :599 */
            if (c) { c[0] = (int)p2x; c[1] = (int)p2y; c[2] = sad; c[3] = bs; }   /* :601 */
        }
        float smooth = calculate_smoothness(i, cand, csize);          /* :605 */
        float t = lambda * (float)mf->lambda_multiplier;              /* :607, left to right */
        t = t * smooth;
        energy[i] = (float)sad + t;
    }
    return min_energy_candidate(energy, csize);                       /* :613 */
}

static void find_min_candidate(orc_mf *mf, orc_level *lv, int pos_x1, int pos_y1,
                               const float (*cand)[2], int csize)
{
    int min_pos = eval_candidates(mf, lv, pos_x1, pos_y1, cand, csize);
    float *f = FLOW_AT(lv, pos_y1, pos_x1);
    f[0] = cand[min_pos][0]; f[1] = cand[min_pos][1];                 /* :616, in place */
}

/* ------------------------------------------------------------------------ */
/* MF::regularize_MVs :424-530.  The nine branches of the reference, in its    */
/* order, each listing its neighbours in the reference's push_back order as    */
/* (row, col) offsets in block units.                                          */
/* ------------------------------------------------------------------------ */
typedef struct { int n; signed char off[9][2]; } nb_case;
static const nb_case NB_INTERIOR = {9, {{0,0},{0,-1},{0,1},{1,1},{-1,-1},{-1,1},{-1,0},{1,0},{1,-1}}}; /* :441-449 */
static const nb_case NB_TOP      = {6, {{0,0},{0,-1},{0,1},{1,1},{1,0},{1,-1}}};                       /* :454-459 */
static const nb_case NB_BOTTOM   = {6, {{0,0},{0,-1},{0,1},{-1,-1},{-1,1},{-1,0}}};                    /* :464-469 */
static const nb_case NB_LEFT     = {6, {{0,0},{0,1},{1,1},{-1,1},{-1,0},{1,0}}};                       /* :474-479 */
static const nb_case NB_RIGHT    = {6, {{0,0},{0,-1},{-1,-1},{-1,0},{1,0},{1,-1}}};                    /* :484-489 */
static const nb_case NB_TL       = {4, {{0,0},{0,1},{1,1},{1,0}}};                                     /* :494-497 */
static const nb_case NB_TR       = {4, {{0,0},{0,-1},{1,0},{1,-1}}};                                   /* :502-505 */
static const nb_case NB_BL       = {4, {{0,0},{0,1},{-1,1},{-1,0}}};                                   /* :510-513 */
static const nb_case NB_BR       = {4, {{0,0},{0,-1},{-1,-1},{-1,0}}};                                 /* :518-521 */

void orc_regularize_mvs(orc_mf *mf, int level)
{
    orc_level *lv = &mf->lv[level];
    int bs = lv->block_size, height = lv->height, width = lv->width;
    if (height / bs < 2 || width / bs < 2) {
        /* fewer than two blocks in a dimension: the reference reads outside the
         * flow field here (:452-522).  Undefined -- the oracle refuses. */
        fprintf(stderr, "orc_regularize_mvs: degenerate %dx%d grid is undefined in the reference\n",
                width / bs, height / bs);
        abort();
    }
    float cand[9][2];
    /* jacobi_regularizer (not the reference; the checker of the product's opt-in fast mode): candidates come from a
     * snapshot of the field taken before the sweep instead of the field being rewritten (:616) */
    int cols = width / bs;
    float *snap = NULL;
    if (mf->jacobi_regularizer) {
        snap = (float *)malloc(sizeof(float) * 2 * (size_t)cols * (height / bs));
        for (int i = 0; i < height; i += bs)
            for (int j = 0; j < width; j += bs) {
                const float *f = FLOW_AT(lv, i, j);
                snap[2 * ((size_t)(i / bs) * cols + j / bs)] = f[0];
                snap[2 * ((size_t)(i / bs) * cols + j / bs) + 1] = f[1];
            }
    }
    for (int i = 0; i < height; i += bs)
        for (int j = 0; j < width; j += bs) {
            const nb_case *c;
            if (i - bs >= 0 && j - bs >= 0 && j + bs < width && i + bs < height) c = &NB_INTERIOR; /* :439 */
            else if (j - bs >= 0 && j + bs < width && i == 0)                    c = &NB_TOP;      /* :452 */
            else if (j - bs >= 0 && j + bs < width && i == height - bs)          c = &NB_BOTTOM;   /* :462 */
            else if (j == 0 && i - bs >= 0 && i + bs < height)                   c = &NB_LEFT;     /* :472 */
            else if (j == width - bs && i - bs >= 0 && i + bs < height)          c = &NB_RIGHT;    /* :482 */
            else if (i == 0 && j == 0)                                           c = &NB_TL;       /* :492 */
            else if (i == 0)                                                     c = &NB_TR;       /* :500 */
            else if (j == 0)                                                     c = &NB_BL;       /* :508 */
            else                                                                 c = &NB_BR;       /* :516 */
            for (int k = 0; k < c->n; ++k) {
                const float *f = snap ? snap + 2 * ((size_t)(i / bs + c->off[k][0]) * cols + j / bs + c->off[k][1])
                                      : FLOW_AT(lv, i + c->off[k][0] * bs, j + c->off[k][1] * bs);
                cand[k][0] = f[0]; cand[k][1] = f[1];
            }
            find_min_candidate(mf, lv, j, i, (const float (*)[2])cand, c->n);   /* :524 */
        }
    free(snap);
}

/* ------------------------------------------------------------------------ */
/* CPU MODEL OF THE GPU SCHEDULE (not a reference function).                   */
/* The raster sweep of regularize_MVs is the unique solution of                */
/*   new[r][c] = F(old[self,R,DR,D,DL], new[L,UL,U,UR])                        */
/* (the dependency graph on `new` is acyclic in raster order).  The GPU solves */
/* it by fixed-point iteration: pass 1 evaluates every block with new := old;  */
/* pass k>1 re-evaluates the blocks one of whose four `new` inputs changed in  */
/* pass k-1; it stops when a pass changes nothing.  This function runs that    */
/* schedule (Jacobi form, two estimate buffers) with the same evaluator as     */
/* the raster sweep so CPU tests can show both give the same field.            */
/* Returns the number of passes run; stats[k] = blocks evaluated in pass k+1.  */
/* ------------------------------------------------------------------------ */
int orc_regularize_fixpoint(orc_mf *mf, int level, int *stats, int max_stats)
{
    orc_level *lv = &mf->lv[level];
    int bs = lv->block_size, rows = lv->height / bs, cols = lv->width / bs;
    size_t n = (size_t)rows * cols;
    float (*oldv)[2] = malloc(sizeof(float[2]) * n);
    float (*est)[2] = malloc(sizeof(float[2]) * n);
    float (*nxt)[2] = malloc(sizeof(float[2]) * n);
    unsigned char *dirty = malloc(n), *ndirty = malloc(n);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            const float *f = FLOW_AT(lv, r * bs, c * bs);
            oldv[(size_t)r * cols + c][0] = f[0]; oldv[(size_t)r * cols + c][1] = f[1];
        }
    memcpy(est, oldv, sizeof(float[2]) * n);
    memset(dirty, 1, n);
    /* candidate order of :441-449; is_new marks inputs already updated in a raster sweep */
    static const signed char off[9][2] = {{0,0},{0,-1},{0,1},{1,1},{-1,-1},{-1,1},{-1,0},{1,0},{1,-1}};
    static const unsigned char is_new[9] = {0, 1, 0, 0, 1, 1, 1, 0, 0};
    int passes = 0;
    for (;;) {
        size_t evaluated = 0, changed = 0;
        memcpy(nxt, est, sizeof(float[2]) * n);
        memset(ndirty, 0, n);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) {
                size_t x = (size_t)r * cols + c;
                if (!dirty[x]) continue;
                ++evaluated;
                float cand[9][2];
                int nc = 0;
                for (int k = 0; k < 9; ++k) {
                    int rr = r + off[k][0], cc = c + off[k][1];
                    if (rr < 0 || rr >= rows || cc < 0 || cc >= cols) continue;
                    const float *s = is_new[k] ? est[(size_t)rr * cols + cc] : oldv[(size_t)rr * cols + cc];
                    cand[nc][0] = s[0]; cand[nc][1] = s[1]; ++nc;
                }
                int mp = eval_candidates(mf, lv, c * bs, r * bs, (const float (*)[2])cand, nc);
                nxt[x][0] = cand[mp][0]; nxt[x][1] = cand[mp][1];
                if (nxt[x][0] != est[x][0] || nxt[x][1] != est[x][1]) {
                    ++changed;
                    /* dependants: the blocks that read this one as a `new` input */
                    static const signed char dep[4][2] = {{0,1},{1,1},{1,0},{1,-1}};
                    for (int d = 0; d < 4; ++d) {
                        int rr = r + dep[d][0], cc = c + dep[d][1];
                        if (rr < 0 || rr >= rows || cc < 0 || cc >= cols) continue;
                        ndirty[(size_t)rr * cols + cc] = 1;
                    }
                }
            }
        if (passes < max_stats && stats) stats[passes] = (int)evaluated;
        ++passes;
        { float (*t)[2] = est; est = nxt; nxt = t; }
        { unsigned char *t = dirty; dirty = ndirty; ndirty = t; }
        if (changed == 0) break;
    }
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            float *f = FLOW_AT(lv, r * bs, c * bs);
            f[0] = est[(size_t)r * cols + c][0]; f[1] = est[(size_t)r * cols + c][1];
        }
    free(oldv); free(est); free(nxt); free(dirty); free(ndirty);
    return passes;
}

/* ------------------------------------------------------------------------ */
/* Level body of MF::calcMotionBlockMatching :115-204 (both branches are the   */
/* same apart from copyMVs), and the tail :205-218.                            */
/* ------------------------------------------------------------------------ */
void orc_level_schedule(orc_mf *mf, int level)
{
    orc_level *lv = &mf->lv[level];
    if (level != mf->num_levels - 1)
        orc_copy_mvs(mf, level);                                      /* :168 */
    orc_calc_level_bm(mf, level);                                     /* :122 / :169 */
    int init_bsize = lv->block_size;                                  /* :133 / :174 */
    float init_lambda = lv->lambda;                                   /* :134 / :175 */
    lv->block_size = init_bsize;
    lv->lambda = init_lambda;
    while (lv->block_size > 1) {                                      /* :141 / :182 */
        for (int l = 0; l < 2; ++l) {                                 /* :143 */
            mf->lambda_multiplier = l + 1;                            /* :145 */
            orc_regularize_mvs(mf, level);                            /* :146 */
        }
        orc_divide_blocks(mf, level);                                 /* :149 */
        lv->block_size = lv->block_size >> 1;                         /* :150 */
        lv->lambda = lv->lambda * 2;                                  /* :151 */
    }
    lv->block_size = init_bsize;                                      /* :154 / :195 */
}

const float *orc_calc_motion_block_matching(orc_mf *mf)
{
    for (int i = mf->num_levels - 1; i >= 0; --i)                     /* :115 */
        orc_level_schedule(mf, i);
    mf->lv[0].block_size = 2;                                         /* :205 */
    orc_copy_to_all_pixels(mf, 0);                                    /* :206 */
    return mf->lv[0].flow;                                            /* :218 */
}

/* ------------------------------------------------------------------------ */
/* Flow: rw_flow.cpp                                                          */
/* ------------------------------------------------------------------------ */
#define ORC_TAG_FLOAT 202021.25f    /* rw_flow.cpp:25 */
#define ORC_TAG_STRING "PIEH"       /* rw_flow.cpp:26 */

static int unknown_flow(float u, float v)                             /* rw_flow.cpp:39-43 */
{
    return (fabs(u) > 1e9) || (fabs(v) > 1e9) || isnan(u) || isnan(v);
}

int orc_flo_read(const char *filename, int *width, int *height, float **data)   /* :50-136 */
{
    if (filename == NULL) return -1;                                  /* :52 */
    const char *dot = strrchr(filename, '.');
    if (dot == NULL || strcmp(dot, ".flo") != 0) return -2;           /* :58-63 (reference derefs NULL) */
    FILE *stream = fopen(filename, "rb");
    if (stream == 0) return -3;                                       /* :66 */
    int w, h; float tag;
    if (fread(&tag, sizeof(float), 1, stream) != 1 ||
        fread(&w, sizeof(int), 1, stream) != 1 ||
        fread(&h, sizeof(int), 1, stream) != 1) { fclose(stream); return -4; }    /* :75-81 */
    if (tag != ORC_TAG_FLOAT) { fclose(stream); return -5; }          /* :82 */
    if (w < 1 || w > 99999) { fclose(stream); return -6; }            /* :88 */
    if (h < 1 || h > 99999) { fclose(stream); return -7; }            /* :94 */
    float *buf = (float *)malloc(sizeof(float) * 2 * (size_t)w * h);
    for (int i = 0; i < h; ++i)                                       /* :106-127 */
        for (int j = 0; j < w; ++j) {
            float *p = buf + 2 * ((size_t)i * w + j);
            if (fread(p, sizeof(float), 1, stream) != 1 ||
                fread(p + 1, sizeof(float), 1, stream) != 1) { free(buf); fclose(stream); return -8; }
        }
    if (fgetc(stream) != EOF) { free(buf); fclose(stream); return -9; }   /* :129 */
    fclose(stream);
    *width = w; *height = h; *data = buf;
    return 0;
}

int orc_flo_write(const char *filename, int width, int height, const float *data)   /* :139-200 */
{
    if (filename == NULL) return -1;                                  /* :141 */
    const char *dot = strrchr(filename, '.');
    if (dot == NULL) return -2;                                       /* :148 */
    if (strcmp(dot, ".flo") != 0) return -3;                          /* :154 */
    FILE *stream = fopen(filename, "wb");
    if (stream == 0) return -4;                                       /* :161 */
    fprintf(stream, ORC_TAG_STRING);                                  /* :168 */
    if (fwrite(&width, sizeof(int), 1, stream) != 1 ||
        fwrite(&height, sizeof(int), 1, stream) != 1) { fclose(stream); return -5; }
    for (int i = 0; i < height; ++i)                                  /* :178-197 */
        for (int j = 0; j < width; ++j) {
            const float *p = data + 2 * ((size_t)i * width + j);
            if (fwrite(p, sizeof(float), 1, stream) != 1 ||
                fwrite(p + 1, sizeof(float), 1, stream) != 1) { fclose(stream); return -6; }
        }
    fclose(stream);
    return 0;
}

double orc_calculate_mse(const float *gtruth, const float *flow, int width, int height)  /* :309-332 */
{
    int count = 0;
    double error = 0;
    for (int i = 0; i < height; ++i)
        for (int j = 0; j < width; ++j) {
            const float *g = gtruth + 2 * ((size_t)i * width + j);
            const float *f = flow + 2 * ((size_t)i * width + j);
            if (unknown_flow(g[0], g[1])) continue;                   /* :318 */
            count++;
            float du = g[0] - f[0], dv = g[1] - f[1];
            float s = du * du + dv * dv;                              /* float expression :325 */
            error += sqrtf(s);                                        /* std::sqrt(float) */
        }
    error = error / count;                                            /* :330 */
    return error;
}

/* ---- Flow::MotionToColor and its helpers (rw_flow.cpp:202-307) ---------------------------- */
static int orc_ncols = 0;
static int orc_colorwheel[60][3];                                     /* MAXCOLS 60, rw_flow.h */

static void orc_setcols(int r, int g, int b, int k)                   /* :302-307 */
{
    orc_colorwheel[k][0] = r; orc_colorwheel[k][1] = g; orc_colorwheel[k][2] = b;
}

static void orc_makecolorwheel(void)                                  /* :277-300 */
{
    const int RY = 15, YG = 6, GC = 4, CB = 11, BM = 13, MR = 6;
    int k = 0;
    orc_ncols = RY + YG + GC + CB + BM + MR;
    for (int i = 0; i < RY; i++) orc_setcols(255, 255 * i / RY, 0, k++);
    for (int i = 0; i < YG; i++) orc_setcols(255 - 255 * i / YG, 255, 0, k++);
    for (int i = 0; i < GC; i++) orc_setcols(0, 255, 255 * i / GC, k++);
    for (int i = 0; i < CB; i++) orc_setcols(0, 255 - 255 * i / CB, 255, k++);
    for (int i = 0; i < BM; i++) orc_setcols(255 * i / BM, 0, 255, k++);
    for (int i = 0; i < MR; i++) orc_setcols(255, 0, 255 - 255 * i / MR, k++);
}

/* :251-275.  The reference is C++: sqrt / atan2 of floats are the float overloads; the division
 * by M_PI and the final 255.0 * col are double expressions; `col *= .75` is a double product
 * rounded back to float. */
/* vendored != 0 selects the expression types of the Middlebury original the reference vendors
 * (middlebury/flow-code/colorcode.cpp:59,65-66: `(a + 1.0) / 2.0 * (ncols-1)` and `/ 255.0` are double
 * there, float `1.0f`, `2.0f`, `255.0f` in rw_flow.cpp:258,264-265).  That flavour exists only so that
 * this restatement can be pinned bit for bit against the compiled vendored file (oracle/_ref). */
static void orc_compute_color(float fx, float fy, unsigned char *pix, int vendored)
{
    if (orc_ncols == 0) orc_makecolorwheel();
    float rad = sqrtf(fx * fx + fy * fy);                             /* :256 */
    float a = (float)((double)atan2f(-fy, -fx) / 3.14159265358979323846);   /* :257 */
    float fk = vendored ? (float)(((double)a + 1.0) / 2.0 * (double)(orc_ncols - 1))
                        : (a + 1.0f) / 2.0f * (float)(orc_ncols - 1); /* :258 */
    int k0 = (int)fk;
    int k1 = (k0 + 1) % orc_ncols;
    float f = fk - (float)k0;
    for (int b = 0; b < 3; b++) {
        float col0 = vendored ? (float)((double)orc_colorwheel[k0][b] / 255.0) : (float)orc_colorwheel[k0][b] / 255.0f;   /* :264 */
        float col1 = vendored ? (float)((double)orc_colorwheel[k1][b] / 255.0) : (float)orc_colorwheel[k1][b] / 255.0f;
        float col = (1 - f) * col0 + f * col1;
        if (rad <= 1) col = 1 - rad * (1 - col);                      /* :268 */
        else col = (float)((double)col * .75);                        /* :270 */
        pix[2 - b] = (unsigned char)(int)(255.0 * (double)col);       /* :271, B,G,R order */
    }
}

void orc_motion_to_color_flavour(const float *flow, int width, int height, float maxmotion,
                                 unsigned char *bgr, float *range, int vendored);

void orc_motion_to_color(const float *flow, int width, int height, float maxmotion,
                         unsigned char *bgr, float *range)            /* :202-249 */
{
    orc_motion_to_color_flavour(flow, width, height, maxmotion, bgr, range, 0);
}

void orc_motion_to_color_flavour(const float *flow, int width, int height, float maxmotion,
                                 unsigned char *bgr, float *range, int vendored)
{
    float maxx = -999, maxy = -999, minx = 999, miny = 999, maxrad = -1;
    for (int i = 0; i < height; i++)
        for (int j = 0; j < width; j++) {
            float fx = flow[2 * ((size_t)i * width + j)], fy = flow[2 * ((size_t)i * width + j) + 1];
            if (unknown_flow(fx, fy)) continue;
            maxx = maxx > fx ? maxx : fx;  maxy = maxy > fy ? maxy : fy;      /* __max / __min :214-217 */
            minx = minx < fx ? minx : fx;  miny = miny < fy ? miny : fy;
            float rad = sqrtf(fx * fx + fy * fy);
            maxrad = maxrad > rad ? maxrad : rad;
        }
    if (range) { range[0] = maxrad; range[1] = minx; range[2] = maxx; range[3] = miny; range[4] = maxy; }   /* printed :223 */
    if (maxmotion > 0) maxrad = maxmotion;                            /* :225 */
    if (maxrad == 0) maxrad = 1;                                      /* :228 */
    for (int i = 0; i < height; i++)
        for (int j = 0; j < width; j++) {
            float fx = flow[2 * ((size_t)i * width + j)], fy = flow[2 * ((size_t)i * width + j) + 1];
            unsigned char *pix = bgr + 3 * ((size_t)i * width + j);
            if (unknown_flow(fx, fy)) pix[0] = pix[1] = pix[2] = 0;   /* :241-243 */
            else orc_compute_color(fx / maxrad, fy / maxrad, pix, vendored);   /* :245 */
        }
}

void orc_subsample_div4(const float *flow_padded, int padded_width, int padded_height,
                        int pad_x, int pad_y, float *out, int out_width, int out_height)
{
    (void)out_height;
    for (int i = pad_y; i < padded_height - pad_y; i += 4)            /* main_class.cpp:63 */
        for (int j = pad_x; j < padded_width - pad_x; j += 4) {       /* :65 */
            const float *f = flow_padded + 2 * ((size_t)i * padded_width + j);
            float *o = out + 2 * ((size_t)((i - pad_y) / 4) * out_width + (j - pad_x) / 4);
            o[0] = f[0] / 4;                                          /* :67 */
            o[1] = f[1] / 4;                                          /* :68 */
        }
}

void orc_free(void *p) { free(p); }
