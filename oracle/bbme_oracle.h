/*
 * bbme_oracle.h -- CPU oracle for the block-matching hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithm (ashish-nr/BlockBasedMotionEstimation, motion_framework.cpp and
 * rw_flow.cpp).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it, and only as the checker.  The product
 * (blockbasedmotionestimation_amd/, libbbme.so) never links or calls it.
 *
 * PARITY STATUS
 *   .flo codec + EPE : pinned.  Checked against the reference's own vendored
 *       Middlebury flowIO.cpp (compiled into oracle/_ref/ by oracle/Makefile)
 *       and against the 8 ground-truth flow10.flo files the reference ships.
 *   search / regulariser / level driver : PARITY UNPINNED.  The reference has
 *       no tests, no golden vectors and no input frames for this path, and its
 *       core (motion_framework.cpp) needs OpenCV 2.4.9/3.0.0, which is absent
 *       here, so it cannot be built.  These functions follow the reference
 *       source line by line (citations on every function) and are
 *       cross-checked by an independent numpy restatement (oracle/bbme_numpy.py).
 *   raster search find_min_block (:246-294, dead code in the reference; raster_search = 1) : PARITY UNPINNED, same
 *       grounds; additionally checked against a direct numpy statement of its rules (tests/test_oracle_cpu.py).
 *   jacobi_regularizer = 1 : NOT a reference function -- the written-down definition of the product's opt-in fast
 *       mode, so that that mode can be tested bit for bit and its distance from the reference's field measured.
 *   padding / pyrDown / 4x bilinear resize : PARITY UNPINNED.  Restated from
 *       OpenCV's published 8-bit algorithms; outside the hot path (host prep).
 *
 * All file:line citations are relative to the reference repository root.
 */
#ifndef BBME_ORACLE_H
#define BBME_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One pyramid level: mirrors PyramidLevel (pyramid_level.h:7-16) plus the
 * per-level SAD cache "fast_array" (motion_framework.h:46). */
typedef struct {
    int width, height;      /* image1.cols / image1.rows                        */
    int block_size;         /* PyramidLevel::block_size (mutated while sweeping) */
    int search_size;        /* PyramidLevel::search_size (window side length)    */
    float lambda;           /* PyramidLevel::lambda                              */
    uint8_t *image1;        /* CV_8UC1, pitch == width                           */
    uint8_t *image2;
    float *flow;            /* level_flow: CV_32FC2 dense, (u,v) interleaved     */
    int32_t *cache;         /* fast_array: CV_32SC4 dense (x2,y2,SAD,bs) or NULL */
} orc_level;

typedef struct {
    int num_levels;
    orc_level *lv;          /* [0] = finest ... [num_levels-1] = coarsest        */
    int lambda_multiplier;  /* MF::lambda_multiplier                             */
    int padded_height, padded_width, padding_x, padding_y; /* public MF fields   */
    int orig_height, orig_width;
    int use_cache;
    int raster_search;     /* 1: calcLevelBM calls find_min_block (:235, commented out in the reference) instead of
                              find_min_block_spiral (:236) */
    int jacobi_regularizer; /* 1: NOT the reference -- every sweep reads only the field as the previous sweep left it (the
                               product's opt-in fast mode); 0: the reference's in-place raster sweep */
} orc_mf;

/* ---- MF constructor pieces (motion_framework.cpp:4-111) ---- */
/* returns 0 ok, -1 "Could not find any multiples..." (reference exits), -2 odd padding difference */
int orc_plan_padding(int width, int height, const int *block_size, int num_levels,
                     int *padded_width, int *padded_height, int *pad_x, int *pad_y);
void orc_pad_zero(const uint8_t *src, int width, int height, int pitch,
                  int pad_x, int pad_y, uint8_t *dst /* (h+2py)*(w+2px) */);
void orc_pyr_down(const uint8_t *src, int sw, int sh, uint8_t *dst /* (sh/2)*(sw/2) */);
void orc_resize_linear_x4(const uint8_t *src, int sw, int sh, uint8_t *dst /* 4sh*4sw */);

int  orc_mf_create(const uint8_t *image1, const uint8_t *image2, int width, int height, int pitch,
                   const int *search_size, const int *block_size, int num_levels,
                   int use_cache, orc_mf **out);
/* build an MF directly from already-made level planes (fixtures taken after the pyramid) */
int  orc_mf_create_from_planes(const uint8_t *const *img1_lv, const uint8_t *const *img2_lv,
                               const int *widths, const int *heights,
                               const int *search_size, const int *block_size, int num_levels,
                               int use_cache, orc_mf **out);
void orc_mf_destroy(orc_mf *mf);

/* ---- hot path, one function per reference method ---- */
void orc_copy_mvs(orc_mf *mf, int level);                 /* MF::copyMVs             :828-843 */
void orc_calc_level_bm(orc_mf *mf, int level);            /* MF::calcLevelBM         :226-244 */
void orc_regularize_mvs(orc_mf *mf, int level);           /* MF::regularize_MVs      :424-530 */
void orc_divide_blocks(orc_mf *mf, int level);            /* MF::divide_blocks       :845-862 */
void orc_copy_to_all_pixels(orc_mf *mf, int level);       /* MF::copy_to_all_pixels  :815-826 */
void orc_level_schedule(orc_mf *mf, int level);           /* body of the level loop  :115-204 */
/* MF::calcMotionBlockMatching :113-219; returns pointer to level 0 flow (padded H0 x W0 x 2) */
const float *orc_calc_motion_block_matching(orc_mf *mf);

/* CPU model of the GPU's fixed-point schedule for one sweep (not a reference function) */
int  orc_regularize_fixpoint(orc_mf *mf, int level, int *stats, int max_stats);

/* single-block entry for unit tests: MF::find_min_block_spiral :296-422 */
void orc_find_min_block_spiral(orc_mf *mf, int level, int image1_ypos, int image1_xpos,
                               int image2_ypos, int image2_xpos, int *pos_x, int *pos_y);
/* literal walk of the spiral for a given shift; writes (dx,dy) per visit, returns count */
int  orc_spiral_walk(int shift, int *dx, int *dy, int cap);

/* ---- Flow (rw_flow.cpp) ---- */
/* ReadFlowFile :50-136. *data is malloc'd (w*h*2 floats); 0 ok, <0 = which check failed */
int    orc_flo_read(const char *filename, int *width, int *height, float **data);
int    orc_flo_write(const char *filename, int width, int height, const float *data); /* :139-200 */
double orc_calculate_mse(const float *gtruth, const float *flow, int width, int height); /* :309-332 */
/* Flow::MotionToColor (rw_flow.cpp:202-307): bgr = width*height*3, range[5] = max radius, min/max u, min/max v */
void   orc_motion_to_color(const float *flow, int width, int height, float maxmotion,
                           unsigned char *bgr, float *range);
/* vendored = 1: expression types of middlebury/flow-code/colorcode.cpp instead of rw_flow.cpp's (pinning only) */
void   orc_motion_to_color_flavour(const float *flow, int width, int height, float maxmotion,
                                   unsigned char *bgr, float *range, int vendored);
/* main_class.cpp:58-70: strip padding, take every 4th pixel, divide by 4 */
void   orc_subsample_div4(const float *flow_padded, int padded_width, int padded_height,
                          int pad_x, int pad_y, float *out, int out_width, int out_height);
void   orc_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
