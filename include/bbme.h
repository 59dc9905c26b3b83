/*
 * bbme.h -- C-ABI of libbbme.so: MI355X-native block-matching motion estimation.
 *
 * Drop-in boundary for the hot path of ashish-nr/BlockBasedMotionEstimation:
 * MF::calcMotionBlockMatching() and everything it calls (pyramidal SAD spiral
 * full search + 8-neighbour MV regularisation), plus the Flow .flo codec and
 * the EPE evaluation.  The reference has no FFI; its boundary is the two C++
 * classes MF (motion_framework.h:9-54) and Flow (rw_flow.h:9-38).  Each entry
 * point below names the reference interface it replaces (paths relative to the
 * reference repository root).  All signatures are plain C: pointers and sizes,
 * no C++ or torch types.  Every function returns 0 (BBME_OK) or a negative
 * bbme_status; bbme_last_error() gives the message (thread-local).  Nothing in
 * this library ever calls exit() or abort().
 *
 * The compute path is HIP on gfx950 only.  There is no CPU fallback: entry
 * points that need the GPU fail with BBME_ERR_HIP when no device is usable.
 */
#ifndef BBME_H
#define BBME_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BBME_MAX_LEVELS 8
#define BBME_MAX_BATCH 64

typedef enum {
    BBME_OK = 0,
    BBME_ERR_INVALID = -1,      /* bad argument (null, non power-of-two block, sizes...)          */
    BBME_ERR_PADDING = -2,      /* "Could not find any multiples of the block size" (:21-26)       */
    BBME_ERR_ODD_PADDING = -3,  /* padded - original is odd: the reference mis-sizes the image     */
    BBME_ERR_DEGENERATE = -4,   /* < 2 blocks in a dimension at some level: reference reads OOB    */
    BBME_ERR_HIP = -5,          /* HIP runtime error / no gfx950 device                            */
    BBME_ERR_IO = -6,           /* .flo file errors (the reference prints and exit(1)s)            */
    BBME_ERR_STATE = -7,        /* call sequence error (e.g. estimate before frames were set)      */
    BBME_ERR_UNSUPPORTED = -8   /* legal in the reference but outside what the kernels implement   */
} bbme_status;

/* Parameters of MF::MF (motion_framework.h:12).  Index 0 = finest level,
 * num_levels-1 = coarsest (motion_framework.cpp:71-72,93-94,115).
 * search_size is the search-window SIDE LENGTH (range = (search_size-block_size)>>1). */
typedef struct {
    int num_levels;
    int block_size[BBME_MAX_LEVELS];
    int search_size[BBME_MAX_LEVELS];
} bbme_params;

typedef struct bbme_ctx bbme_ctx;

/* ---- no GPU needed ---------------------------------------------------------------- */

const char *bbme_version(void);
const char *bbme_last_error(void);

/* Padding search of MF::MF (motion_framework.cpp:14-54): public fields
 * padded_width/padded_height/padding_x/padding_y (motion_framework.h:16-19). */
int bbme_plan_padding(int width, int height, const bbme_params *params,
                      int *padded_width, int *padded_height, int *pad_x, int *pad_y);

/* Host-side pieces of MF::MF, exposed for callers that build planes themselves:
 * zero border (copyMakeBorder, :60-61) and one pyrDown step (:89-90). */
int bbme_pad_zero_host(const uint8_t *src, int width, int height, int pitch,
                       int pad_x, int pad_y, uint8_t *dst);
int bbme_pyr_down_host(const uint8_t *src, int src_width, int src_height, uint8_t *dst);
/* cv::resize(..., 4, 4, INTER_LINEAR) of main_class.cpp:32-33 (8-bit fixed point). */
int bbme_resize_x4_host(const uint8_t *src, int src_width, int src_height, uint8_t *dst);

/* Flow::ReadFlowFile (rw_flow.cpp:50-136).  *data is allocated by the library
 * (width*height*2 floats, u,v interleaved, row-major); release with bbme_free. */
int bbme_flo_read(const char *filename, int *width, int *height, float **data);
/* Flow::WriteFlowFile (rw_flow.cpp:139-200). */
int bbme_flo_write(const char *filename, int width, int height, const float *data);
/* Flow::WriteFlowFile on a worker thread (the reference's writer, rw_flow.cpp:139-200, has no caller and
 * is synchronous): submit hands over `height` rows of `width` (u, v) pairs starting at `data`, consecutive rows
 * `pitch_pixels` pixels apart -- e.g. the unpadded window of the padded field bbme_get_flow_host left in pinned memory:
 * data = flow + 2 * (pad_y * padded_width + pad_x), pitch_pixels = padded_width (main_class.cpp:63-70) -- and returns at
 * once; a writer with one worker writes the files in submission order, byte for byte what bbme_flo_write produces.  The
 * memory must stay untouched until a wait that covers the job returns; waits also report the first I/O error since the
 * last one.
 * bbme_flo_writer_create_pool: `workers` threads, one file each at a time -- files are independent, and one 66 MB file is
 * bound by the kernel's write path (three writers in flight write a 4K sequence twice as fast as
 * one); files then finish in any order.  Every submitted job has a ticket, 1, 2, ... in submission order:
 * bbme_flo_writer_ticket gives the last one handed out, bbme_flo_writer_wait_ticket returns once every job up to that
 * ticket is on disk (bbme_flo_writer_wait: every job submitted so far) -- what a pipeline needs that re-uses a staging
 * buffer round by round (csrc/seq_schedule.hpp: only round k - 2 has to be on disk before round k's download). */
typedef struct bbme_flo_writer bbme_flo_writer;
int bbme_flo_writer_create(bbme_flo_writer **out);
int bbme_flo_writer_create_pool(int workers, bbme_flo_writer **out);
int bbme_flo_writer_ticket(bbme_flo_writer *w, unsigned long long *ticket);
int bbme_flo_writer_wait_ticket(bbme_flo_writer *w, unsigned long long ticket);
int bbme_flo_writer_submit(bbme_flo_writer *w, const char *filename, int width, int height, const float *data,
                           int pitch_pixels);
/* The same file from the compact result (bbme_get_cells_host: one int16 (dx, dy) pair per 2x2 pixels of the padded level-0
 * field, cell_rows x cell_cols): copy_to_all_pixels (motion_framework.cpp:815-826), the padding strip (main_class.cpp:63-70)
 * and WriteFlowFile's rows fused on the worker, which expands and writes bands of rows with a few helper threads
 * (BBME_WRITER_THREADS, default 2; the kernel's write path is the bound).  Only 1/16 of the dense field crosses PCIe, and nobody holds 8 bytes per pixel in
 * memory.  Pixel (x, y) of the file = cell ((y + pad_y) / 2, (x + pad_x) / 2).  Byte for byte the file bbme_flo_write makes
 * of the dense field's window.  `cells` must stay untouched until bbme_flo_writer_wait returns. */
int bbme_flo_writer_submit_cells(bbme_flo_writer *w, const char *filename, int width, int height, const int16_t *cells,
                                 int cell_rows, int cell_cols, int pad_x, int pad_y);
int bbme_flo_writer_wait(bbme_flo_writer *w);
int bbme_flo_writer_destroy(bbme_flo_writer *w);
/* Flow::CalculateMSE (rw_flow.cpp:309-332): mean end-point error over known GT pixels. */
int bbme_calculate_mse(const float *gtruth, const float *flow, int width, int height, double *out);
/* Flow::MotionToColor (rw_flow.cpp:202-249, with computeColor :251-275 and makecolorwheel :277-300):
 * Middlebury colour coding of a flow field.  bgr = width*height*3 bytes, B,G,R per pixel as in the
 * reference's CV_8UC3 image; unknown pixels black; maxmotion > 0 overrides the normalising radius.
 * range (may be NULL) receives {max radius, min u, max u, min v, max v} of the known pixels, the
 * numbers the reference prints. */
int bbme_motion_to_color(const float *flow, int width, int height, float maxmotion, uint8_t *bgr, float *range);
/* What Flow::ShowImage (rw_flow.cpp:334-340) keeps on disk, as binary PPM (no PNG codec, no GUI here). */
int bbme_ppm_write_bgr(const char *filename, int width, int height, const uint8_t *bgr);
/* main_class.cpp:58-70: strip padding, every 4th pixel, divide by 4. */
int bbme_subsample_div4(const float *flow_padded, int padded_width, int padded_height,
                        int pad_x, int pad_y, float *out, int out_width, int out_height);
void bbme_free(void *p);

/* Host tables of the search kernels, exposed for tests.  bbme_spiral_host: visiting order of
 * find_min_block_spiral (motion_framework.cpp:326-411), rank -> (dx, dy).  bbme_search_plan_host: the
 * work split of k_search_fast -- per round a code rounds[] = S | kind << 8 and 64 tasks (0xffffffff = idle lane):
 *   kind 0  strips: a lane takes column group g (candidate columns 4g .. 4g+3) and the S candidate rows from dy0;
 *           task = g | dy0 << 8;
 *   kind 1  the last candidate row of the tight plan (even ranges, block <= 16): the four lanes of a quad share one
 *           (group, row), a quarter of the block's rows each; task = g | dy << 8 | part << 16, part = 0..3;
 *   kind 2  the last candidate column of the tight plan (dx = +R), one candidate per lane; task = dy.
 * Together the tasks cover every candidate of the (2R+1)^2 square exactly once (kind 1: once per part). */
int bbme_spiral_host(int search_size, int block_size, int16_t *dx, int16_t *dy, int capacity, int *count);
int bbme_search_plan_host(int range, int block_size, uint32_t *rounds, int rounds_capacity, int *nrounds,
                          uint32_t *tasks /* rounds_capacity * 64 */, int *groups, int *pitch_dw);
/* the split used when `waves` (1 or 2) waves share a macroblock (levels with fewer blocks than the chip has SIMDs):
 * 64 * waves tasks per round, wave w takes tasks [64 w, 64 w + 64) of each */
int bbme_search_plan_host_waves(int range, int block_size, int waves, uint32_t *rounds, int rounds_capacity, int *nrounds,
                                uint32_t *tasks /* rounds_capacity * 64 * waves */, int *groups, int *pitch_dw);

/* ---- context: one per GPU stream (replaces an MF object) ---------------------------- */

/* Allocates every device buffer for a (width x height) frame pair: padded planes of
 * all levels, MV grids, work lists, the dense output.  device = HIP ordinal. */
int bbme_create(const bbme_params *params, int width, int height, int device, bbme_ctx **out);
/* A context for `pairs` (1..BBME_MAX_BATCH) independent frame pairs of one size: `pairs` MF objects behind one launch
 * sequence.  The reference holds all state of a pair in one MF object and carries nothing from pair to pair
 * (motion_framework.h:37-46), so the pairs of a sequence can be estimated side by side: every kernel of bbme_estimate then
 * works on all pairs at once (one more grid dimension), which is how a sequence keeps one GPU busy -- the regulariser of a
 * single pair is a chain of short dependent launches that leaves most of the chip idle, and the device dispatches dependent
 * kernels of many streams no faster than one per few microseconds.  Each pair's field is bit for bit what a context of its
 * own would produce.  Entry points without a pair index address pair 0. */
int bbme_create_batch(const bbme_params *params, int width, int height, int device, int pairs, bbme_ctx **out);
int bbme_batch_size(const bbme_ctx *ctx, int *pairs);
int bbme_destroy(bbme_ctx *ctx);
/* hipStream_t to run on (default: a stream the ctx creates).  Pass the raw handle. */
int bbme_set_stream(bbme_ctx *ctx, void *hip_stream);
/* The raw hipStream_t the context enqueues on (to order other work, e.g. a collective, behind bbme_estimate). */
int bbme_get_stream(bbme_ctx *ctx, void **hip_stream);
int bbme_get_geometry(const bbme_ctx *ctx, int *padded_width, int *padded_height,
                      int *pad_x, int *pad_y);
int bbme_level_geometry(const bbme_ctx *ctx, int level, int *width, int *height,
                        int *block_size, int *search_size);

/* ---- inputs ------------------------------------------------------------------------- */

/* MF::MF(image1, image2, ...) (motion_framework.cpp:4-111) for host images: uploads the two frames as they are and runs
 * the zero border (:57-61) and the pyrDown cascade (:86-106) as HIP kernels on the ctx stream, exactly as
 * bbme_set_frames_device does; returns when the upload has completed (the caller may re-use its buffers). */
int bbme_set_frames_host(bbme_ctx *ctx, const uint8_t *image1, const uint8_t *image2, int pitch);
int bbme_set_frames_host_pair(bbme_ctx *ctx, int pair, const uint8_t *image1, const uint8_t *image2, int pitch);
/* The same without the host wait: upload, border and pyramid are only enqueued on the ctx stream (truly asynchronous when
 * the source buffers are pinned: hipHostMalloc / hipHostRegister).  The buffers must stay untouched until the ctx stream
 * has passed this point (bbme_synchronize, or an event the caller records on bbme_get_stream's stream). */
int bbme_set_frames_host_async(bbme_ctx *ctx, int pair, const uint8_t *image1, const uint8_t *image2, int pitch);
/* Same constructor for frames already resident in HBM (unpadded, width x height):
 * zero padding and the whole pyrDown cascade run as HIP kernels on the ctx stream. */
int bbme_set_frames_device(bbme_ctx *ctx, const uint8_t *d_image1, const uint8_t *d_image2, int pitch);
int bbme_set_frames_device_pair(bbme_ctx *ctx, int pair, const uint8_t *d_image1, const uint8_t *d_image2, int pitch);
/* Which of the reference's two block searches MF::calcLevelBM calls (motion_framework.cpp:235-236): the spiral full
 * search find_min_block_spiral (:296-422, the live one: ties go to the candidate visited first on the spiral; a
 * prediction outside the image gives a zero MV) or the raster full search find_min_block (:246-294, commented out in the
 * reference: window clamped to the image, ties go to the candidate closer (L1) to the block's own position, then to the
 * first in raster order; a window entirely outside the image leaves the prediction as the result).  Default: spiral. */
enum { BBME_SEARCH_SPIRAL = 0, BBME_SEARCH_RASTER = 1 };
int bbme_set_search_mode(bbme_ctx *ctx, int mode);
/* BBME_REG_EXACT (default): every sweep leaves exactly the field of the reference's in-place raster sweep
 * (regularize_MVs writes each winner straight back, :616, and later blocks of the same sweep read it, :441-449).
 * BBME_REG_JACOBI: opt-in fast mode, NOT the reference's result -- every block of a sweep is evaluated against the field
 * as the previous sweep left it (one fully parallel pass per sweep, no dependent chains).  Same candidates, energies and
 * tie rules per block; the fields differ where a sweep's changes would have propagated within the sweep.  bench.py
 * reports its speed and its end-point error beside the exact mode's, never as `value`. */
enum { BBME_REG_EXACT = 0, BBME_REG_JACOBI = 1 };
int bbme_set_regularizer_mode(bbme_ctx *ctx, int mode);
/* Scheduling option (default on; BBME_SPECULATE=0 turns the default off): bbme_estimate starts the search of every level
 * but the coarsest on a second stream beside the coarser level's late regulariser sweeps, predicting from that level's grid
 * as it stands, and afterwards searches again the blocks whose prediction those sweeps changed.  Same field, bit for bit;
 * shorter single pairs (the late sweeps leave most of the chip idle).  Turn it off when several contexts keep the chip
 * busy anyway (sequences with pairs in flight). */
int bbme_set_speculation(bbme_ctx *ctx, int enabled);
/* Scheduling option (default on): the relaxation launch (k_reg_iter) in front of the solver on large grids of small blocks takes
 * the heavy first generations of a sweep off the solver's latency-bound waves at the price of chip-wide work.  Same field,
 * bit for bit.  Turn it off, like the speculation, when several contexts keep the chip busy (8 pairs in flight at 4K:
 * 31.6 -> 33.2 Mblocks/s). */
int bbme_set_relaxation(bbme_ctx *ctx, int enabled);
/* Orders the ctx stream behind everything enqueued so far on another HIP stream of the same device (NULL = the default
 * stream): call it before bbme_set_frames_device when the frames were produced by asynchronous work on that stream.
 * Without it the caller must have synchronised the producer itself. */
int bbme_wait_for_stream(bbme_ctx *ctx, void *producer_stream);
/* Direct access to the ctx-owned padded planes of a level (device pointers, pitch ==
 * level width) so a caller can fill or inspect them in place.
 * SINGLE-PAIR ENTRY POINTS.  The three plane calls below, bbme_calculate_mse_device, every bbme_stage_* call,
 * bbme_sweep_stats, bbme_last_sweep_passes and bbme_gather_cells (bbme_rccl.h) address one pair: on a batched context
 * (bbme_create_batch with pairs > 1) they return BBME_ERR_UNSUPPORTED instead of quietly working on pair 0.  A batch
 * is fed with bbme_set_frames_{host,device}_pair and read with the *_pair getters. */
int bbme_level_planes_device(bbme_ctx *ctx, int level, uint8_t **d_image1, uint8_t **d_image2);
/* Upload ready-made padded planes of one level (fixtures taken after the pyramid). */
int bbme_set_level_planes_host(bbme_ctx *ctx, int level, const uint8_t *image1, const uint8_t *image2);
int bbme_get_level_planes_host(bbme_ctx *ctx, int level, uint8_t *image1, uint8_t *image2);

/* ---- the hot path ------------------------------------------------------------------- */

/* MF::calcMotionBlockMatching() (motion_framework.cpp:113-219).  Enqueues the whole
 * pyramid (search + regularisation of every level + dense expansion) on the ctx
 * stream and returns without waiting; no host synchronisation inside. */
int bbme_estimate(bbme_ctx *ctx);
int bbme_synchronize(bbme_ctx *ctx);
/* The cv::Mat returned by calcMotionBlockMatching (:218): dense padded H0 x W0
 * float2 (u,v) = (dx,dy), device pointer, pitch == padded width. */
int bbme_flow_device(bbme_ctx *ctx, const float **d_flow);
int bbme_flow_device_pair(bbme_ctx *ctx, int pair, const float **d_flow);
/* Synchronises, then copies the dense padded field to the host. */
int bbme_get_flow_host(bbme_ctx *ctx, float *flow /* padded_h * padded_w * 2 */);
int bbme_get_flow_host_pair(bbme_ctx *ctx, int pair, float *flow);
/* Compact result: one int16 (dx,dy) pair per 2x2 cell of level 0 ((H0/2) x (W0/2)). */
int bbme_cells_device(bbme_ctx *ctx, const int16_t **d_cells);
int bbme_cells_device_pair(bbme_ctx *ctx, int pair, const int16_t **d_cells);
/* copy_to_all_pixels (:815-826) for a cell grid that lives anywhere in HBM (e.g. gathered from
 * another GPU): writes the dense padded H0 x W0 float2 field to d_flow, on the ctx stream. */
int bbme_expand_cells_device(bbme_ctx *ctx, const int16_t *d_cells, float *d_flow);
/* The same on a caller-supplied HIP stream (NULL = the ctx stream), e.g. the stream a gather completes on,
 * so that the expansion of one step's results overlaps the next step's estimate. */
int bbme_expand_cells_device_on(bbme_ctx *ctx, const int16_t *d_cells, float *d_flow, void *hip_stream);
int bbme_get_cells_host(bbme_ctx *ctx, int16_t *cells);
int bbme_get_cells_host_pair(bbme_ctx *ctx, int pair, int16_t *cells);
/* Flow::CalculateMSE (rw_flow.cpp:309-332) on the device, fused with the driver's subsampling
 * (main_class.cpp:58-70): mean end-point error between a ground-truth field in HBM (gt_width x gt_height,
 * u,v interleaved) and the context's current result taken at every `scale`-th pixel of the unpadded frame and
 * divided by `scale` (4 for the reference's pipeline, 1 for none).  Per-pixel arithmetic is the reference's
 * float expression; the double sum is taken in a different order, so it agrees with bbme_calculate_mse to
 * about 1e-12 relative, not bit for bit.  Synchronises the ctx stream. */
int bbme_calculate_mse_device(bbme_ctx *ctx, const float *d_gtruth, int gt_width, int gt_height, int scale, double *out);

/* ---- single stages, for parity tests against the reference's private methods (single-pair contexts only) -------- */

/* copyMVs (:828-843) + calcLevelBM (:226-244) of one level.  Leaves that level's MV
 * grid at block size block_size[level]. */
int bbme_stage_search(bbme_ctx *ctx, int level);
/* One regularize_MVs() sweep (:424-530) at block size `block` with lambda_multiplier
 * `mult` (lambda follows the reference's rule lambda = (B/2) * (B/block)).  `block` must
 * equal the level's current grid block size, or half of it (then divide_blocks, :845-862,
 * is applied first). */
int bbme_stage_regularize(bbme_ctx *ctx, int level, int block, int mult);
/* Current MV grid of a level, sampled at `block` (<= current grid block size), as
 * int16 (dx,dy) per block, row-major (height/block rows x width/block cols). */
int bbme_stage_get_mvs(bbme_ctx *ctx, int level, int block, int16_t *mvs);
/* Overwrite a level's MV grid at block size `block` (makes it the current grid). */
int bbme_stage_set_mvs(bbme_ctx *ctx, int level, int block, const int16_t *mvs);
/* copy_to_all_pixels (:815-826) on level 0 after its last divide: fills the dense field. */
int bbme_stage_expand(bbme_ctx *ctx);
/* Raw counters of the last sweep (diagnostic, 16 words): [3] safety-net passes, [4] blocks
 * re-evaluated by the solver, [5] non-convergence flag, [7] most rounds run by one wave,
 * [8] rounds summed over waves; sweeps with the SAD memo (block >= 8): [9] candidate look-ups
 * of the chain rounds, [10] of them not in the memo, [11] group passes that summed those,
 * [12] changes whose dependants' SADs were forwarded; [13] (rounds << 16 | rounds that left
 * queued blocks waiting) of the wave with the most rounds, [14] such rounds summed over waves
 * (how much of a sweep is queueing rather than a dependency chain).  [4] and [7..14] are only counted by
 * sweeps run through bbme_stage_regularize (hundreds of waves adding to the same words is a
 * queue at the memory side that bbme_estimate does not stand in); [3] and [5] always. */
int bbme_sweep_stats(bbme_ctx *ctx, unsigned *stats16);
/* Fix-up passes the last sweep needed after its first full pass (diagnostic). */
int bbme_last_sweep_passes(bbme_ctx *ctx, int *passes);

/* Times (ms, HIP events on the ctx stream) of the last bbme_estimate when profiling
 * was enabled: total, and the sum over levels of search / regulariser / expand kernels. */
int bbme_set_profiling(bbme_ctx *ctx, int enabled);
int bbme_get_timings(bbme_ctx *ctx, float *total_ms, float *search_ms, float *regularize_ms,
                     float *expand_ms, float *search_level0_ms);

/* Measures the chip-wide issue rate of v_qsad_pk_u16_u8 (gops[0]) and v_sad_u8 (gops[1]) in 1e9
 * wave-instructions per second (8 waves per SIMD, 8 independent chains per lane): the VALU
 * ceilings bench.py prices the search kernel against.  gops[2] / gops[3]: QSAD rate when every QSAD
 * is interleaved with one / four independent v_sad_u8 (do the two instructions overlap?). */
int bbme_probe_rates(int device, double *gops);

/* The two candidate inner loops of the search kernel, reduced to their LDS reads and SAD instructions and run at the
 * occupancy their LDS footprints allow: tabs[0] = v_qsad_pk_u16_u8 strips over one copy of the window (what k_search_fast
 * does, 6.75 KB per wave), tabs[1] = v_sad_u8 over four byte-shifted copies (27 KB per wave), in 1e12 abs-diff per second.
 * The measurement behind the choice of instruction. */
int bbme_probe_search_loops(int device, double *tabs2);

/* Dependent-chain latency of the memory operations a solver round is made of, one lane on an idle
 * chip: out[2k] = shader cycles per operation, out[2k+1] = 10 ns ticks for 256 operations, for
 * k = 0 plain load, 1 agent-scope load, 2 returning atomic, 3 agent-scope store + drain. */
int bbme_probe_latency(int device, unsigned long long *out9);

/* Checks on the device what the kernels' XCD-aware block orders assume: that workgroups b and b + 8 of a launch run on
 * the same XCD (HW_REG_XCC_ID read by 4096 workgroups).  xcds_seen = distinct XCDs, violations = workgroups whose XCD
 * differs from that of workgroup b mod 8.  Speed only -- no result depends on the placement. */
int bbme_probe_xcd(int device, int *xcds_seen, int *violations);

/* Profiling aid: launches a kernel that reads `mbytes` MiB exactly once with one aligned dword per
 * lane (the access shape of the search kernel's window staging), `repeats` times, so that the
 * FETCH_SIZE counter can be calibrated against a known byte count in the same rocprofv3 run. */
int bbme_calibrate_read(int device, unsigned mbytes, int repeats);

/* Instruction probes used by the GPU test-suite: checks v_sad_u8, v_alignbyte_b32,
 * v_qsad_pk_u16_u8 and v_sad_u16 against a scalar model on 65536 random operands, and unaligned
 * dword / dwordx2 / dwordx4 global loads against byte-assembled values.
 * mismatches[0..4] receive the number of disagreements per item, in that order. */
int bbme_selftest_isa(int device, int *mismatches);

#ifdef __cplusplus
}
#endif
#endif /* BBME_H */
