// bbme_device.hip -- context, launch sequence and the GPU half of the C-ABI (include/bbme.h).
// Replaces the MF object of the reference (motion_framework.h:9-54): bbme_create + bbme_set_frames_*
// are MF::MF, bbme_estimate is MF::calcMotionBlockMatching.  gfx950 only; no CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bbme_internal.hpp"
#include "bbme_kernels.hpp"

using namespace bbme;

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return bbme::fail(BBME_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                              __FILE__, __LINE__);                                             \
    } while (0)

namespace {

struct Level {
    int width = 0, height = 0, block = 0, search = 0, range = 0;
    uint8_t *img1 = nullptr, *img2 = nullptr;     // padded planes, pitch == width
    // MV grids.  small[]: grids at the level's own block size B (the search writes small[0]; the two sweeps at B go
    // small[0] -> small[1] -> small[0], which then stays untouched until the level's next search: the speculative search
    // of the next finer level predicts from it).  big[]: grids at b < B (capacity (H/2)*(W/2)), ping-pong.
    mv_t *small[2] = {nullptr, nullptr};
    mv_t *big[2] = {nullptr, nullptr};
    mv_t *cur_grid = nullptr;                     // the grid that holds the current field
    int cur_block = 0;                            // its block size (0 = nothing yet)
    mv_t *pred = nullptr;                         // per block: the coarse MV a speculative search started from
    uint32_t *fix_list = nullptr, *fix_count = nullptr;   // blocks to search again after a speculative search
    mv_t *final_grid() const { return block == 2 ? small[0] : big[1]; }   // where two sweeps per block size leave the 2x2 cells
    uint32_t *spiral = nullptr;                   // rank -> packed (dx, dy)
    int ncand = 0;
    int pitch_dw = 0;
    size_t lds_bytes = 0;
    // fast search kernel (block 8 / 16)
    bool fast = false;
    uint16_t *rank_of = nullptr;
    int rank_pitch = 0;
    uint32_t *tasks = nullptr, *rounds = nullptr;
    int nrounds = 0;
    uint32_t *tasks2 = nullptr, *rounds2 = nullptr;   // the plan for two waves per macroblock (levels of few blocks)
    int nrounds2 = 0;
    uint2 *lane_ranks = nullptr, *lane_ranks2 = nullptr;   // per plan: the ranks of every lane's candidates (FastSearchArgs::lane_ranks)
    bool split_pays = false;                          // the two-wave plan is at least 20 % shorter per wave
    int fast_pitch_dw = 0;
    size_t fast_lds_bytes = 0;
    // batch: every per-pair buffer holds ctx->batch copies, pair p at p * stride elements (multiples of 64 elements)
    uint32_t plane_stride = 0;                        // img1 / img2, bytes
    uint32_t small_stride = 0;                        // small[], pred, fix_list: words
    uint32_t big_stride = 0;                          // big[]: words
    uint32_t grid_stride(const mv_t *g) const { return (g == small[0] || g == small[1]) ? small_stride : big_stride; }
};

}  // namespace

struct bbme_ctx {
    bbme_params params{};
    Geometry geom{};
    int device = 0;
    int batch = 1;                                // independent frame pairs this context holds (blockIdx.y of every kernel)
    size_t flow_stride = 0;                       // floats from pair to pair in `flow`
    uint32_t list_stride = 0, own_stride = 0;     // words from pair to pair in list[] / own
    size_t raw_stride = 0;                        // bytes from pair to pair in raw[]
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::vector<Level> lv;
    float *flow = nullptr;                        // dense padded H0 x W0 float2
    uint32_t *list[2] = {nullptr, nullptr};
    uint8_t *flags[2] = {nullptr, nullptr};       // dirty flags of the regulariser, one byte per block, all zero between sweeps
    size_t flag_bytes = 0;
    int relax_steps = -1;                         // k_reg_iter launches per sweep; -1 = by grid size (BBME_RELAX_STEPS overrides)
    bool split_forced = false;                    // threshold given in the environment: split whatever the plans' lengths (tests)
    long long scan_fine_max = 140000;             // grids of at most this many blocks: scan segments of 4 flags (BBME_SCAN_FINE_MAX)
    long long pass1_lanes_max = 140000;           // grids of at most this many blocks: pass 1 in the chain form (BBME_PASS1_LANES_MAX)
    bool list_split = true;                       // two waves per listed block in the fix-up search of small levels; BBME_LIST_SPLIT
    bool pass1_lazy = true;                       // pass 1 leaves its evaluations to a relaxation launch that follows it; BBME_PASS1_LAZY
    int pass1_strip = -1;                         // the strip form of pass 1 at b <= 4 (k_reg_pass1_strip): -1 = batched contexts only; BBME_PASS1_STRIP
    int split_blocks = 10000;                     // levels of at most this many macroblocks: two waves per block (BBME_SEARCH_SPLIT_BLOCKS)
    int round_cap = 0;                            // > 0: test knob, the regulariser's waves give up after this many rounds
    uint32_t *own = nullptr;                      // ownership counters of the solver, one word per block
    uint32_t own_pitch = 0;                       // transposed layout: 32 residue classes of own_pitch words
    uint32_t *counters = nullptr;                 // 64 words (RegArgs::counters)
    // SAD memo of the regulariser's chain form (bbme_kernels.hpp, "SAD memo"): nine (MV, SAD) words per block at b >= 8
    unsigned long long *memo = nullptr;
    uint32_t memo_stride = 0;                     // words from pair to pair
    size_t memo_blocks = 0;                       // blocks per pair it has room for
    int memo_level = -1, memo_block = 0;          // the (level, block size) its slots describe; block 0 = nothing
    bool use_memo = true;                         // BBME_MEMO
    int memo_min_block = 16;                      // sweeps at smaller blocks run without it (8: measured slower); BBME_MEMO_MIN_B
    bool memo_forward = false;                    // BBME_MEMO_FORWARD (measured slower: off)
    uint64_t frames_mask = 0;                     // bit p: pair p has frames (bbme_estimate needs every pair's)
    bool frames_set() const { return frames_mask == (batch >= 64 ? ~0ull : (1ull << batch) - 1ull); }
    double *epe_scratch = nullptr;                // partial sums + counts of bbme_calculate_mse_device (allocated on first use)
    int local_rounds = 8;                         // k_reg_iter: heavy rounds of a tile per launch; BBME_LOCAL_ROUNDS
    int wide_threshold = 16;                      // solver: queue length above which a round takes the throughput form; BBME_WIDE_THRESHOLD
    int solve_waves = 4;                          // waves per solver workgroup (1, 2 or 4); BBME_SOLVE_WAVES
    bool solve_share = true;                      // k_reg_solve: idle waves of a workgroup take a sibling's surplus; BBME_SOLVE_SHARE=0
    int solve_wgs = 256;                          // most workgroups of k_reg_solve (4 independent waves each); r04: 256 measured 1.5 % ahead of 128 (one wave per SIMD)
    int xcd_remap = 1;                            // XCD-aware block order in k_search_fast; BBME_XCD_REMAP
    bool jacobi = false;                          // opt-in, not bit-exact: Jacobi sweeps (pass 1 only); bbme_set_regularizer_mode
    bool raster_search = false;                   // MF::find_min_block (:246-294) instead of the spiral search; bbme_set_search_mode
    bool force_generic_search = false;            // BBME_GENERIC_SEARCH=1: use k_search_generic everywhere
    bool use_graph = true;
    bool relax = true;                            // relaxation launches (k_reg_iter) on large grids of small blocks; bbme_set_relaxation
    bool speculate = true;                        // overlap every level's search with the coarser level's late sweeps; BBME_SPECULATE
    hipStream_t side_stream = nullptr;            // the speculative searches
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int spec_per_cu = 0;                          // BBME_SPEC_WGS_PER_CU; 0 = by the coarser level's block size (spec_lds_for)
    // LDS per workgroup of a speculative search launch beside the late sweeps of `coarser_block`-sized level: the occupancy cap
    int spec_per_cu_l0 = 0;                       // second value of BBME_SPEC_WGS_PER_CU="other,level0": the level-0 launch's own cap
    size_t spec_lds_for(int coarser_block, int level = 1) const
    {
        // (r04: 7, not 6, since the level-0 search -- not the level-1 sweeps beside it, shorter now -- is what the level waits for:
        //  cfg3 1.494 -> 1.470 ms; 8 and more cost the sweeps more than the search gains)
        int per_cu = spec_per_cu > 0 ? spec_per_cu : (coarser_block >= 16 ? 7 : 24);
        if (level == 0 && spec_per_cu_l0 > 0) per_cu = spec_per_cu_l0;
        return ((size_t)(160 - 40) * 1024 / per_cu) / 256 * 256;
    }
    double spec_min_absdiffs = 8e9;               // levels with less search work are not speculated; BBME_SPEC_MIN_GABS
    hipGraphExec_t graph_exec = nullptr;
    bool profiling = false;
    float t_total = 0, t_search = 0, t_reg = 0, t_expand = 0, t_search0 = 0;
    uint8_t *raw[2] = {nullptr, nullptr};         // bbme_set_frames_host: the unpadded frames in HBM (allocated on first use)
};

namespace {

int check_ctx(const bbme_ctx *c)
{
    if (!c) return bbme::fail(BBME_ERR_INVALID, "ctx is null");
    return BBME_OK;
}

// The stage-by-stage entry points, the plane injection, the sweep counters, the device-side EPE and the RCCL gather address ONE
// pair: on a batched context (bbme_create_batch with pairs > 1) they are refused rather than silently applied to pair 0.
int single_pair_only(const bbme_ctx *c, const char *what)
{
    if (int rc = check_ctx(c)) return rc;
    if (c->batch > 1)
        return bbme::fail(BBME_ERR_UNSUPPORTED, "%s addresses one pair: not available on a batched context (%d pairs)", what, c->batch);
    return BBME_OK;
}

int check_level(const bbme_ctx *c, int level)
{
    if (int rc = check_ctx(c)) return rc;
    if (level < 0 || level >= (int)c->lv.size())
        return bbme::fail(BBME_ERR_INVALID, "level %d out of range 0..%d", level, (int)c->lv.size() - 1);
    return BBME_OK;
}

void drop_graph(bbme_ctx *c)
{
    if (c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
}

// The regulariser's waves leave at a round cap instead of spinning for ever (RegArgs::round_cap); a sweep that hit
// it has not reached the fixed point and its field must not be handed out as a result.  Waits for the stream.  On
// that path the solver's ownership words and counters are stale too: cleared, so that the context stays usable.
int check_converged(bbme_ctx *c)
{
    std::vector<uint32_t> flags((size_t)c->batch, 0u);    // counters[5] of every pair (64 words apart)
    HIP_TRY(hipMemcpy2DAsync(flags.data(), sizeof(uint32_t), c->counters + 5, 64 * sizeof(uint32_t), sizeof(uint32_t),
                             (size_t)c->batch, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    bool flag = false;
    for (uint32_t f : flags) flag = flag || f != 0;
    if (!flag) return BBME_OK;
    HIP_TRY(hipMemsetAsync(c->own, 0, (size_t)c->own_stride * 4 * c->batch, c->stream));
    HIP_TRY(hipMemsetAsync(c->counters, 0, (size_t)256 * c->batch, c->stream));
    HIP_TRY(hipMemsetAsync(c->flags[0], 0, c->flag_bytes * c->batch, c->stream));
    HIP_TRY(hipMemsetAsync(c->flags[1], 0, c->flag_bytes * c->batch, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return bbme::fail(BBME_ERR_STATE, "a regulariser sweep hit its round cap without converging: the motion field is not "
                                      "the reference's and has been discarded");
}

// ---- launches ---------------------------------------------------------------------------

template <int B>
void launch_search_t(const SearchArgs &a, int nblocks, int batch, size_t lds, hipStream_t s)
{
    hipLaunchKernelGGL(k_search_generic<B>, dim3(nblocks, batch), dim3(64), lds, s, a);
}

// Where the search of `level` takes its predictions from (copyMVs, :828-843), by mode:
//   plain / fix-up : the coarser level's final 2x2-cell grid (it must have been regularised down to 2x2);
//   speculative    : the coarser level's grid as its two sweeps at its own block size left it (Level::small[0]).
template <class Args>
int set_prediction_source(bbme_ctx *c, int level, int mode, Args &a)
{
    Level &L = c->lv[level];
    a.mode = mode;
    a.pred = L.pred;
    a.coarse = nullptr;
    a.s_plane = L.plane_stride; a.s_pred = L.small_stride; a.s_out = L.small_stride; a.s_coarse = 0;
    if (level + 1 >= (int)c->lv.size()) return BBME_OK;
    Level &C = c->lv[level + 1];
    a.coarse_block = C.block;
    if (mode == kSearchSpeculative) {
        if (C.cur_block != C.block || C.cur_grid != C.small[0])
            return bbme::fail(BBME_ERR_STATE, "level %d is not at the end of the sweeps at its own block size", level + 1);
        a.coarse = C.small[0];
        a.coarse_cell_shift = 0;
        while ((1 << a.coarse_cell_shift) < C.block) ++a.coarse_cell_shift;
    } else {
        if (C.cur_block != 2)
            return bbme::fail(BBME_ERR_STATE, "level %d has not been regularised down to 2x2 blocks", level + 1);
        a.coarse = C.cur_grid;
        a.coarse_cell_shift = 1;
    }
    a.coarse_cols = C.width >> a.coarse_cell_shift;
    a.s_coarse = C.grid_stride(a.coarse);
    return BBME_OK;
}

// `lds_floor`: dynamic LDS to ask for at least -- the speculative launch pads its workgroups so that only
// ctx::spec_wgs_per_cu of them fit a CU and the regulariser's kernels beside it still find wave slots, registers and LDS.
int launch_search_fast(bbme_ctx *c, int level, int mode, hipStream_t stream, size_t lds_floor)
{
    Level &L = c->lv[level];
    FastSearchArgs a{};
    a.image1 = L.img1; a.image2 = L.img2;
    a.width = L.width; a.height = L.height;
    a.range = L.range; a.spiral = L.spiral;
    a.rank_of = L.rank_of; a.rank_pitch = L.rank_pitch;
    a.tasks = L.tasks; a.rounds = L.rounds; a.nrounds = L.nrounds; a.lane_ranks = L.lane_ranks;
    {
        const uint32_t nch = (uint32_t)(L.fast_pitch_dw + 3) / 4;
        a.stage_magic = (65536u + nch - 1) / nch;
        a.stage_rpp = 64u / nch;
    }
    if (int rc = set_prediction_source(c, level, mode, a)) return rc;
    a.out = L.small[0];
    a.cols = L.width / L.block;
    a.pitch_dw = L.fast_pitch_dw;
    const int nblocks = (L.width / L.block) * (L.height / L.block);
    a.nblocks = nblocks;
    a.cols_magic = (uint64_t)nblocks * (uint64_t)a.cols < (1ull << 32) ? (uint32_t)(((1ull << 32) + (uint64_t)a.cols - 1) / (uint64_t)a.cols) : 0u;
    a.xcd_remap = c->xcd_remap;
    const int grid = c->xcd_remap ? ((nblocks + 7) / 8) * 8 : nblocks;
    const size_t lds = std::max(L.fast_lds_bytes, lds_floor);
    a.fix_count = L.fix_count;
    a.s_fix_list = L.small_stride;
    const unsigned P = (unsigned)c->batch;
    // a level with fewer macroblocks than the chip has SIMDs: one wave per block leaves most SIMDs idle and every busy one
    // with a single wave, so the launch lasts as long as one block does -- two waves share each block then
    if (mode == kSearchPlain && L.tasks2 && nblocks <= c->split_blocks && (L.split_pays || c->split_forced)) {
        a.tasks = L.tasks2; a.rounds = L.rounds2; a.nrounds = L.nrounds2; a.lane_ranks = L.lane_ranks2;
        a.stage_rpp = 128u / ((uint32_t)(L.fast_pitch_dw + 3) / 4);
        if (L.block == 16) hipLaunchKernelGGL((k_search_fast<16, 2>), dim3(grid, P), dim3(128), lds, stream, a);
        else if (L.block == 32) hipLaunchKernelGGL((k_search_fast<32, 2>), dim3(grid, P), dim3(128), lds, stream, a);
        else hipLaunchKernelGGL((k_search_fast<8, 2>), dim3(grid, P), dim3(128), lds, stream, a);
        HIP_TRY(hipGetLastError());
        return BBME_OK;
    }
    if (mode == kSearchFixup && a.coarse) {
        // list the blocks whose prediction changed, then search those (k_fixup_list, k_search_list)
        a.mode = kSearchPlain;
        hipLaunchKernelGGL(k_fixup_list, dim3((nblocks + 255) / 256, P), dim3(256), 0, stream, a, L.block, L.fix_count, L.fix_list);
        const int lgrid = std::max(64, nblocks / 4);
        // two waves per listed block on the levels that are searched with two waves per block anyway (r04): the list is ONE
        // generation of waves, and on a level of 8 160 blocks (~1 200 listed) halves fill the chip where wholes leave three SIMDs
        // in four idle -- 30.4 -> 24.8 us.  Level 0 (~5 000 listed) stays with one wave per block: at two, the 123 registers of
        // the 128-lane form allow four waves per SIMD, 10 000 halves are two and a half generations, 58.8 -> 65.8 us.
        if (c->list_split && L.tasks2 && nblocks <= c->split_blocks && (L.split_pays || c->split_forced)) {
            a.tasks = L.tasks2; a.rounds = L.rounds2; a.nrounds = L.nrounds2; a.lane_ranks = L.lane_ranks2;
            a.stage_rpp = 128u / ((uint32_t)(L.fast_pitch_dw + 3) / 4);
            if (L.block == 16) hipLaunchKernelGGL((k_search_list<16, 2>), dim3(lgrid, P), dim3(128), lds, stream, a, L.fix_count, L.fix_list);
            else if (L.block == 32) hipLaunchKernelGGL((k_search_list<32, 2>), dim3(lgrid, P), dim3(128), lds, stream, a, L.fix_count, L.fix_list);
            else hipLaunchKernelGGL((k_search_list<8, 2>), dim3(lgrid, P), dim3(128), lds, stream, a, L.fix_count, L.fix_list);
            HIP_TRY(hipGetLastError());
            return BBME_OK;
        }
        if (L.block == 16) hipLaunchKernelGGL(k_search_list<16>, dim3(lgrid, P), dim3(64), lds, stream, a, L.fix_count, L.fix_list);
        else if (L.block == 32) hipLaunchKernelGGL(k_search_list<32>, dim3(lgrid, P), dim3(64), lds, stream, a, L.fix_count, L.fix_list);
        else hipLaunchKernelGGL(k_search_list<8>, dim3(lgrid, P), dim3(64), lds, stream, a, L.fix_count, L.fix_list);
        HIP_TRY(hipGetLastError());
        return BBME_OK;
    }
    if (L.block == 16)
        hipLaunchKernelGGL((k_search_fast<16, 1>), dim3(grid, P), dim3(64), lds, stream, a);
    else if (L.block == 32)
        hipLaunchKernelGGL((k_search_fast<32, 1>), dim3(grid, P), dim3(64), lds, stream, a);
    else
        hipLaunchKernelGGL((k_search_fast<8, 1>), dim3(grid, P), dim3(64), lds, stream, a);
    HIP_TRY(hipGetLastError());
    return BBME_OK;
}

int launch_search(bbme_ctx *c, int level, int mode = kSearchPlain, hipStream_t stream = nullptr, size_t lds_floor = 0)
{
    Level &L = c->lv[level];
    if (!stream) stream = c->stream;
    int rc;
    if (L.fast && !c->force_generic_search && !c->raster_search) {
        rc = launch_search_fast(c, level, mode, stream, lds_floor);
    } else {
        SearchArgs a{};
        a.image1 = L.img1; a.image2 = L.img2;
        a.width = L.width; a.height = L.height;
        a.range = L.range; a.ncand = L.ncand; a.spiral = L.spiral;
        a.raster = c->raster_search ? 1 : 0;
        if ((rc = set_prediction_source(c, level, mode, a))) return rc;
        a.out = L.small[0];
        a.cols = L.width / L.block;
        a.pitch_dw = L.pitch_dw;
        const int nblocks = (L.width / L.block) * (L.height / L.block);
        const size_t lds = std::max(L.lds_bytes, lds_floor);
        switch (L.block) {
        case 2:  launch_search_t<2>(a, nblocks, c->batch, lds, stream); break;
        case 4:  launch_search_t<4>(a, nblocks, c->batch, lds, stream); break;
        case 8:  launch_search_t<8>(a, nblocks, c->batch, lds, stream); break;
        case 16: launch_search_t<16>(a, nblocks, c->batch, lds, stream); break;
        case 32: launch_search_t<32>(a, nblocks, c->batch, lds, stream); break;
        case 64: launch_search_t<64>(a, nblocks, c->batch, lds, stream); break;
        default: return bbme::fail(BBME_ERR_UNSUPPORTED, "block size %d", L.block);
        }
        HIP_TRY(hipGetLastError());
        rc = BBME_OK;
    }
    if (rc == BBME_OK && mode != kSearchSpeculative) { L.cur_grid = L.small[0]; L.cur_block = L.block; }
    return rc;
}

template <int BS>
void launch_sweep_t(RegArgs a, uint8_t *const flags[2], int relax_steps, int max_solve_wgs, int solve_waves, bool jacobi,
                    long long lanes_max, long long fine_max, int strip, bool lazy_ok, unsigned P, hipStream_t s)
{
    constexpr int LPB = RegCfg<BS>::LPB;
    const long long blocks = (long long)a.rows * a.cols;
    const int grid1 = (int)((blocks * LPB + 255) / 256);
    // a multiple of 8 workgroups: one share per XCD (k_reg_solve's bands)
    const int grid2 = (int)((std::min<long long>(max_solve_wgs, (blocks + 63) / 64) + 7) / 8 * 8);
    // pass 1 marks flags[0]; relaxation step i consumes flags[i & 1] and marks the other; the solver
    // consumes what the last step marked.  Every flag is zero again afterwards.
    // grids up to ~130 000 blocks: the chain form of pass 1 (a third of the instructions per wave; 16 lanes per block fill the
    // chip from ~32 000 blocks on, and it still wins up to four times that: 1.815 -> 1.78 ms per cfg3 pair; slower from 500 000)
    auto pass1 = [&]() {
        if constexpr (BS >= 8) {
            if (a.memo) {                                  // (launch_sweep hands a memo only to sweeps whose pass 1 has the chain form)
                hipLaunchKernelGGL((k_reg_pass1_lanes<BS, true>), dim3((unsigned)((blocks * 16 + 255) / 256), P), dim3(256), 0, s, a);
                return;
            }
        }
        if (BS <= 16 && blocks <= lanes_max)               // (b >= 32: a lane would walk 32+ rows -- 13 against 6 us at b = 32)
            hipLaunchKernelGGL(k_reg_pass1_lanes<BS>, dim3((unsigned)((blocks * 16 + 255) / 256), P), dim3(256), 0, s, a);
        else {
            if constexpr (BS <= 4) {
                // the large grids of small blocks in a BATCHED context: strips of four blocks per lane, the blocks that need their
                // images listed per wave and worked off densely.  A third fewer vector instructions per pair -- which a batch,
                // throughput-bound, turns into time (24 pairs as 4 x 6: 60.0 -> 62.3 Mblocks/s) -- in a quarter of the waves, each of
                // which now walks its list pass after pass -- which a single pair, latency-bound, pays for (level 0, b = 4: 15.8 ->
                // 53 us; 1.563 -> 1.657 ms per step).  BBME_PASS1_STRIP=0 / 1 forces it off / on.
                // ... and, for any context, the sweeps with a relaxation launch behind pass 1: the strip test alone (level 0: 5.1 us at
                // b = 4, 6.5 at b = 2, against 15.5 / 18.4 for the whole of k_reg_pass1), the blocks that need their images flagged for
                // the relaxation's first round (RegArgs::lazy)
                a.lazy = (lazy_ok && relax_steps > 0 && a.flag_next != nullptr) ? 1 : 0;
                const bool strip_form = a.lazy || (strip < 0 ? P > 1 : strip != 0);
                if (strip_form && a.cols % 4 == 0 && a.cols >= 12) {
                    hipLaunchKernelGGL(k_reg_pass1_strip<BS>, dim3((unsigned)((blocks / 4 + 255) / 256), P), dim3(256), 0, s, a);
                    return;
                }
            }
            a.lazy = 0;
            hipLaunchKernelGGL(k_reg_pass1<BS>, dim3(grid1, P), dim3(256), 0, s, a);
        }
    };
    if (jacobi) {
        // opt-in, NOT the reference's field: every block against the field as the previous sweep left it, and no more
        a.flag_cur = nullptr; a.flag_next = nullptr;
        pass1();
        return;
    }
    a.flag_cur = nullptr; a.flag_next = flags[0];
    pass1();
    int cur = 0;
    for (int i = 0; i < relax_steps; ++i, cur ^= 1) {
        a.flag_cur = flags[cur]; a.flag_next = flags[cur ^ 1];
        constexpr int T = RegIter<BS>::T;
        const unsigned tiles = (unsigned)(((a.cols + T - 1) / T) * ((a.rows + T - 1) / T));
        hipLaunchKernelGGL(k_reg_iter<BS>, dim3(tiles, P), dim3(256), 0, s, a);
    }
    a.flag_cur = flags[cur]; a.flag_next = nullptr;
    // small grids: scan segments of 4 flags, so that the stale blocks of a row are dealt to four times as many waves
    a.memo_init = 0;                                       // pass 1 has written every slot
    if constexpr (BS >= 8) {
        if (a.memo) {
            if (blocks <= fine_max) hipLaunchKernelGGL((k_reg_solve<BS, 4, true>), dim3(grid2, P), dim3(64 * solve_waves), 0, s, a);
            else hipLaunchKernelGGL((k_reg_solve<BS, 16, true>), dim3(grid2, P), dim3(64 * solve_waves), 0, s, a);
            return;
        }
    }
    if (blocks <= fine_max) hipLaunchKernelGGL((k_reg_solve<BS, 4>), dim3(grid2, P), dim3(64 * solve_waves), 0, s, a);
    else hipLaunchKernelGGL((k_reg_solve<BS, 16>), dim3(grid2, P), dim3(64 * solve_waves), 0, s, a);
}

int launch_sweep(bbme_ctx *c, int level, int b, int mult, bool stats = false)
{
    Level &L = c->lv[level];
    if (mult < 1) return bbme::fail(BBME_ERR_INVALID, "lambda multiplier %d", mult);
    if (b < 2 || b > L.block || (b & (b - 1)))
        return bbme::fail(BBME_ERR_INVALID, "block %d is not a power of two in 2..%d", b, L.block);
    RegArgs a{};
    if (L.cur_block == b) a.old_shift = 0;
    else if (L.cur_block == 2 * b) a.old_shift = 1;
    else return bbme::fail(BBME_ERR_STATE, "level %d grid is at block size %d, cannot sweep at %d",
                           level, L.cur_block, b);
    a.image1 = L.img1; a.image2 = L.img2;
    a.width = L.width; a.height = L.height;
    a.rows = L.height / b; a.cols = L.width / b;
    a.old_grid = L.cur_grid;
    a.old_cols = a.cols >> a.old_shift;
    // sweeps at the level's own block size ping-pong in small[], the others in big[] (see Level)
    mv_t *const *pool = (b == L.block) ? L.small : L.big;
    a.est = (L.cur_grid == pool[0]) ? pool[1] : pool[0];
    // lambda = (float)(B/2), doubled at every halving (motion_framework.cpp:73,95,151); times
    // (float)lambda_multiplier as at :607
    float lambda = (float)(L.block / 2);
    for (int s = L.block; s > b; s >>= 1) lambda = lambda * 2;
    a.lambda_mult = lambda * (float)mult;
    a.list0 = c->list[0]; a.list1 = c->list[1];
    a.own = c->own;
    a.own_pitch = c->own_pitch;
    a.s_plane = L.plane_stride; a.s_old = L.grid_stride(a.old_grid); a.s_est = L.grid_stride(a.est);
    a.s_list = c->list_stride; a.s_own = c->own_stride; a.s_flag = (uint32_t)c->flag_bytes;
    a.local_rounds = c->local_rounds;
    a.wide_threshold = (uint32_t)c->wide_threshold;
    // every round of a wave either empties part of its queue or follows a real change, and a change can only travel
    // along the raster dependency chain (< 2 * rows + cols blocks): the cap is an exit every wave reaches even if
    // that reasoning were wrong; hitting it raises counters[5] and the result is refused (BBME_ERR_STATE)
    a.round_cap = c->round_cap > 0 ? (uint32_t)c->round_cap : 64u * (uint32_t)(2 * a.rows + a.cols + 16);
    a.counters = c->counters;
    a.stats = stats ? 1 : 0;
    a.share = c->solve_share ? 1 : 0;
    // the SAD memo: sweeps at b >= 8 whose pass 1 runs in the chain form (it is what fills the slots); the first sweep at a
    // (level, block size) finds nothing in it and rewrites every slot
    const long long nblk_memo = (long long)a.rows * a.cols;
    if (c->use_memo && c->memo && !c->jacobi && b >= c->memo_min_block && nblk_memo <= c->pass1_lanes_max && (size_t)nblk_memo <= c->memo_blocks &&
        L.width <= 8192 && L.height <= 8192) {             // (group_sads packs a vector into 2 x 14 bits)
        a.memo = c->memo;
        a.s_memo = c->memo_stride;
        a.memo_init = !(c->memo_level == level && c->memo_block == b);
        a.memo_forward = c->memo_forward ? 1 : 0;
        c->memo_level = level; c->memo_block = b;
    }
    // relaxation launches (k_reg_iter, 8 local rounds per tile): one more launch (>= 5 us), which only the sweeps with
    // heavy first generations repay -- measured on cfg3 / cfg4 / cfg2: large grids of small blocks, one launch per sweep
    const long long nblk = (long long)a.rows * a.cols;
    int steps = c->relax_steps;
    if (steps < 0) {
        // BBME_RELAX_RULE="min_blocks,max_b,steps_first,steps_second" (tuning knob)
        // (r03: no relaxation launch in front of the second sweep at a block size -- it changes little, and the launch cost more
        // than it took off the solver: 1.760 -> 1.735 ms per cfg3 pair; r04: nor at 4 x 4 -- with the memo-less solver of this
        // round the chain form takes those sweeps' first generations faster than a 40 us launch does: cfg3 1.566 -> 1.547 ms,
        // cfg4 1.605 -> 1.585 ms, interleaved medians of 5 / 4 runs; and, once the solver's waves shared their queues, only on
        // grids of >= 300 000 blocks: cfg3 1.523 -> 1.496, cfg2 0.699 -> 0.680, cfg4 1.611 -> 1.603, reference literals 1.336 -> 1.331)
        static long long min_blocks = 300000;
        static int max_b = 2, s1 = 1, s2 = 0;
        static const bool parsed = [] {
            if (const char *e = getenv("BBME_RELAX_RULE")) sscanf(e, "%lld,%d,%d,%d", &min_blocks, &max_b, &s1, &s2);
            return true;
        }();
        (void)parsed;
        steps = (c->relax && nblk >= min_blocks && b <= max_b) ? (mult == 1 ? s1 : s2) : 0;
    }
    switch (b) {
    case 2:  launch_sweep_t<2>(a, c->flags, steps, c->solve_wgs, c->solve_waves, c->jacobi, c->pass1_lanes_max, c->scan_fine_max, c->pass1_strip, c->pass1_lazy, (unsigned)c->batch, c->stream); break;
    case 4:  launch_sweep_t<4>(a, c->flags, steps, c->solve_wgs, c->solve_waves, c->jacobi, c->pass1_lanes_max, c->scan_fine_max, c->pass1_strip, c->pass1_lazy, (unsigned)c->batch, c->stream); break;
    case 8:  launch_sweep_t<8>(a, c->flags, steps, c->solve_wgs, c->solve_waves, c->jacobi, c->pass1_lanes_max, c->scan_fine_max, c->pass1_strip, c->pass1_lazy, (unsigned)c->batch, c->stream); break;
    case 16: launch_sweep_t<16>(a, c->flags, steps, c->solve_wgs, c->solve_waves, c->jacobi, c->pass1_lanes_max, c->scan_fine_max, c->pass1_strip, c->pass1_lazy, (unsigned)c->batch, c->stream); break;
    case 32: launch_sweep_t<32>(a, c->flags, steps, c->solve_wgs, c->solve_waves, c->jacobi, c->pass1_lanes_max, c->scan_fine_max, c->pass1_strip, c->pass1_lazy, (unsigned)c->batch, c->stream); break;
    case 64: launch_sweep_t<64>(a, c->flags, steps, c->solve_wgs, c->solve_waves, c->jacobi, c->pass1_lanes_max, c->scan_fine_max, c->pass1_strip, c->pass1_lazy, (unsigned)c->batch, c->stream); break;
    default: return bbme::fail(BBME_ERR_UNSUPPORTED, "block size %d", b);
    }
    HIP_TRY(hipGetLastError());
    L.cur_grid = a.est;
    L.cur_block = b;
    return BBME_OK;
}

int launch_expand(bbme_ctx *c)
{
    Level &L = c->lv[0];
    if (L.cur_block != 2) return bbme::fail(BBME_ERR_STATE, "level 0 has not been regularised down to 2x2 blocks");
    const int cc = L.width / 2, cr = L.height / 2;
    const long long threads = (long long)cc * cr * 2;
    hipLaunchKernelGGL(k_expand, dim3((unsigned)((threads + 255) / 256), (unsigned)c->batch), dim3(256), 0, c->stream,
                       L.cur_grid, cc, cr, c->flow, L.width, L.grid_stride(L.cur_grid), c->flow_stride);
    HIP_TRY(hipGetLastError());
    return BBME_OK;
}

// A speculative search pays its fork, its fix-up launches and the stretch it puts on the sweeps beside it only when the
// search it hides is long: levels below ~8 G abs-diffs (~90 us) are searched in line (measured: cfg2, cfg1 and the reference's
// literals lose 3-12 % when every level is speculated; cfg3 / cfg4 gain 8-10 % from their two largest levels).
bool worth_speculating(const bbme_ctx *c, int level)
{
    if (c->jacobi) return false;                     // Jacobi sweeps are too short to hide a search behind
    const Level &L = c->lv[level];
    const double side = 2.0 * L.range + 1.0;
    const double absdiffs = (double)(L.width / L.block) * (L.height / L.block) * side * side * L.block * L.block;
    return absdiffs >= c->spec_min_absdiffs;
}

// The level loop of MF::calcMotionBlockMatching (:115-206).  With `speculate`, the search of level l-1 is started on a
// second stream as soon as level l has finished the two sweeps at its own block size, and runs beside the level's
// remaining sweeps (which are latency-bound and leave most of the chip idle); when the level is final, a fix-up launch
// searches again the blocks whose prediction those sweeps changed (search_prediction, bbme_kernels.hpp).
int enqueue_pyramid(bbme_ctx *c, bool speculate)
{
    const int nl = (int)c->lv.size();
    if (speculate && nl > 1 && !c->side_stream) {       // only contexts that speculate hold a second stream (hardware queue)
        // lowest dispatch priority: the regulariser's workgroups on the main stream go first whenever both have some ready
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, lo));
    }
    bool speculated = false;
    c->memo_block = 0;                                  // a captured launch sequence must not depend on what ran before it
    for (int l = nl - 1; l >= 0; --l) {
        if (speculated) {
            HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
            if (int rc = launch_search(c, l, kSearchFixup)) return rc;
        } else if (int rc = launch_search(c, l)) return rc;
        speculated = false;
        for (int b = c->lv[l].block; b > 1; b >>= 1) {            // while (block_size > 1) :141
            for (int mult = 1; mult <= 2; ++mult)                  // lambda_multiplier = l + 1 :145
                if (int rc = launch_sweep(c, l, b, mult)) return rc;
            if (speculate && l > 0 && b == c->lv[l].block && b > 2 && worth_speculating(c, l - 1)) {
                HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
                HIP_TRY(hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
                if (int rc = launch_search(c, l - 1, kSearchSpeculative, c->side_stream, c->spec_lds_for(c->lv[l].block, l - 1))) return rc;
                HIP_TRY(hipEventRecord(c->ev_join, c->side_stream));
                speculated = true;
            }
        }
    }
    return launch_expand(c);
}

int profiled_pyramid(bbme_ctx *c)
{
    // eager launches with events between sections (rank-0 diagnostics; not the timed bench path)
    std::vector<hipEvent_t> ev;
    auto mark = [&]() -> int {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        HIP_TRY(hipEventRecord(e, c->stream));
        ev.push_back(e);
        return BBME_OK;
    };
    std::vector<int> kind;   // 0 search, 1 reg, 2 expand ; section i lies between ev[i] and ev[i+1]
    std::vector<int> lvl;
    if (int rc = mark()) return rc;
    c->memo_block = 0;
    for (int l = (int)c->lv.size() - 1; l >= 0; --l) {
        if (int rc = launch_search(c, l)) return rc;
        if (int rc = mark()) return rc;
        kind.push_back(0); lvl.push_back(l);
        for (int b = c->lv[l].block; b > 1; b >>= 1)
            for (int mult = 1; mult <= 2; ++mult)
                if (int rc = launch_sweep(c, l, b, mult)) return rc;
        if (int rc = mark()) return rc;
        kind.push_back(1); lvl.push_back(l);
    }
    if (int rc = launch_expand(c)) return rc;
    if (int rc = mark()) return rc;
    kind.push_back(2); lvl.push_back(0);
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->t_search = c->t_reg = c->t_expand = c->t_search0 = 0;
    for (size_t i = 0; i < kind.size(); ++i) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
        if (kind[i] == 0) { c->t_search += ms; if (lvl[i] == 0) c->t_search0 = ms; }
        else if (kind[i] == 1) c->t_reg += ms;
        else c->t_expand += ms;
    }
    HIP_TRY(hipEventElapsedTime(&c->t_total, ev.front(), ev.back()));
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    return BBME_OK;
}

}  // namespace

// =========================================================================================
// C-ABI
// =========================================================================================
extern "C" {

int bbme_create(const bbme_params *params, int width, int height, int device, bbme_ctx **out)
{
    return bbme_create_batch(params, width, height, device, 1, out);
}

int bbme_create_batch(const bbme_params *params, int width, int height, int device, int pairs, bbme_ctx **out)
{
    if (!params || !out) return bbme::fail(BBME_ERR_INVALID, "bbme_create: null argument");
    *out = nullptr;
    if (pairs < 1 || pairs > BBME_MAX_BATCH) return bbme::fail(BBME_ERR_INVALID, "batch of %d pairs (1..%d)", pairs, BBME_MAX_BATCH);
    if (int rc = validate_params(*params)) return rc;
    Geometry g;
    if (int rc = plan_padding(width, height, *params, g)) return rc;
    const int nl = params->num_levels;
    // grids with fewer than two blocks in a dimension make regularize_MVs read outside the
    // flow field in the reference (motion_framework.cpp:452-522): undefined there, refused here
    for (int l = 0; l < nl; ++l) {
        const int w = g.padded_width >> l, h = g.padded_height >> l, b = params->block_size[l];
        if (w / b < 2 || h / b < 2)
            return bbme::fail(BBME_ERR_DEGENERATE,
                              "level %d is %dx%d with %dx%d blocks: fewer than two blocks in a dimension "
                              "is undefined behaviour in the reference", l, w, h, b, b);
    }
    // the kernels move level rows as dwords (pitch == level width): with blocks of 4 x 4 and more every level width is a multiple
    // of four by construction, with 2 x 2 blocks it need not be
    for (int l = 0; l < nl; ++l)
        if ((g.padded_width >> l) % 4 != 0)
            return bbme::fail(BBME_ERR_UNSUPPORTED, "level %d is %d pixels wide: the kernels need level widths that are multiples of 4 "
                                                     "(2x2 blocks on a frame this narrow)", l, g.padded_width >> l);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return bbme::fail(BBME_ERR_HIP, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return bbme::fail(BBME_ERR_INVALID, "device %d of %d", device, ndev);
    HIP_TRY(hipSetDevice(device));

    bbme_ctx *c = new bbme_ctx();
    c->params = *params; c->geom = g; c->device = device;
    c->batch = pairs;
    const size_t P = (size_t)pairs;
    auto round64 = [](size_t n) { return (n + 63) / 64 * 64; };
    if (const char *e = getenv("BBME_SOLVE_SHARE")) c->solve_share = atoi(e) != 0;
    if (const char *e = getenv("BBME_SOLVE_WGS")) c->solve_wgs = std::max(1, std::min(8192, atoi(e)));
    if (const char *e = getenv("BBME_RELAX_STEPS")) c->relax_steps = std::max(0, std::min(64, atoi(e)));
    if (const char *e = getenv("BBME_SOLVE_WAVES")) { const int v = atoi(e); c->solve_waves = v <= 1 ? 1 : (v == 2 ? 2 : 4); }
    if (const char *e = getenv("BBME_TEST_ROUND_CAP")) c->round_cap = std::max(0, atoi(e));
    // a batched context is throughput-bound (every launch carries several pairs): the chain form of pass 1, which trades
    // instructions for latency, only pays on its small grids (24 pairs as 4 x 6: 53.7 -> 55.0 Mblocks/s)
    if (pairs > 1) c->pass1_lanes_max = 40000;
    if (const char *e = getenv("BBME_PASS1_LANES_MAX")) c->pass1_lanes_max = atoll(e);
    if (const char *e = getenv("BBME_SCAN_FINE_MAX")) c->scan_fine_max = atoll(e);
    if (const char *e = getenv("BBME_PASS1_STRIP")) c->pass1_strip = atoi(e) != 0 ? 1 : 0;
    if (const char *e = getenv("BBME_LIST_SPLIT")) c->list_split = atoi(e) != 0;
    if (const char *e = getenv("BBME_PASS1_LAZY")) c->pass1_lazy = atoi(e) != 0;
    if (const char *e = getenv("BBME_SEARCH_SPLIT_BLOCKS")) { c->split_blocks = std::max(0, atoi(e)); c->split_forced = true; }
    if (const char *e = getenv("BBME_NO_GRAPH")) c->use_graph = atoi(e) == 0;
    if (const char *e = getenv("BBME_SPECULATE")) c->speculate = atoi(e) != 0;
    {
        // a speculative search may keep at most this many of its (one-wave) workgroups on a CU: the rest of the CU's wave
        // slots, registers and LDS (40 KB) stay free for the regulariser kernels it runs beside
        // (r04, on the faster solver.  Behind a level of 16 x 16 blocks -- three late block sizes to hide the search behind -- 6-7 is
        // best: cfg3 1.614 / 1.585 / 1.591 / 1.626 ms at 8 / 6 / 7 / 5.  Behind a level of 8 x 8 blocks the sweeps are over long
        // before the search is, and any cap only delays it: cfg4 1.80 / 1.72 / 1.67 / 1.62 ms at 6 / 8 / 10 / 24 = uncapped.)
        if (const char *e = getenv("BBME_SPEC_WGS_PER_CU")) {
            c->spec_per_cu = std::max(1, std::min(32, atoi(e)));
            if (const char *comma = strchr(e, ',')) c->spec_per_cu_l0 = std::max(1, std::min(32, atoi(comma + 1)));
        }
        if (const char *e = getenv("BBME_SPEC_MIN_GABS")) c->spec_min_absdiffs = atof(e) * 1e9;
    }
    if (const char *e = getenv("BBME_GENERIC_SEARCH")) c->force_generic_search = atoi(e) != 0;
    if (const char *e = getenv("BBME_LOCAL_ROUNDS")) c->local_rounds = std::max(1, atoi(e));
    if (const char *e = getenv("BBME_WIDE_THRESHOLD")) c->wide_threshold = std::max(4, atoi(e));
    if (const char *e = getenv("BBME_MEMO")) c->use_memo = atoi(e) != 0;
    if (const char *e = getenv("BBME_MEMO_FORWARD")) c->memo_forward = atoi(e) != 0;
    if (const char *e = getenv("BBME_MEMO_MIN_B")) c->memo_min_block = std::max(8, atoi(e));
    if (const char *e = getenv("BBME_XCD_REMAP")) c->xcd_remap = atoi(e) != 0;
    c->lv.resize(nl);
    auto cleanup_fail = [&](int rc) { bbme_destroy(c); return rc; };
    hipError_t err = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (err != hipSuccess) return cleanup_fail(bbme::fail(BBME_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(err)));
    c->own_stream = true;
    if ((err = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming)) != hipSuccess ||
        (err = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming)) != hipSuccess)
        return cleanup_fail(bbme::fail(BBME_ERR_HIP, "creating the side stream: %s", hipGetErrorString(err)));
    size_t max_blocks = 0;
    for (int l = 0; l < nl; ++l) {
        Level &L = c->lv[l];
        L.width = g.padded_width >> l; L.height = g.padded_height >> l;
        L.block = params->block_size[l]; L.search = params->search_size[l];
        // the memo serves the sweeps at b >= memo_min_block whose grid the chain-form pass 1 takes: room for the largest of them
        for (int b = c->memo_min_block; b <= L.block; b <<= 1) {
            const size_t nb = (size_t)(L.width / b) * (L.height / b);
            if ((long long)nb <= c->pass1_lanes_max) { c->memo_blocks = std::max(c->memo_blocks, nb); break; }
        }
        SpiralTable sp = build_spiral(L.search, L.block);
        L.range = sp.range; L.ncand = (int)sp.dx.size();
        L.pitch_dw = (L.block + 2 * L.range) / 4 + 2;
        L.lds_bytes = ((size_t)(L.block + 2 * L.range) * L.pitch_dw + (size_t)L.block * L.block / 4) * 4;
        const size_t plane = round64((size_t)L.width * L.height + 64) + 192;   // slack: row_sad may touch 3 bytes past the end
        const size_t cells = round64((size_t)(L.width / 2) * (L.height / 2));
        const size_t own_blocks = round64((size_t)(L.width / L.block) * (L.height / L.block));
        L.plane_stride = (uint32_t)plane; L.small_stride = (uint32_t)own_blocks; L.big_stride = (uint32_t)cells;
        max_blocks = std::max(max_blocks, cells);
        std::vector<uint32_t> packed(sp.dx.size());
        for (size_t i = 0; i < sp.dx.size(); ++i)
            packed[i] = ((uint32_t)(uint16_t)sp.dx[i]) | ((uint32_t)(uint16_t)sp.dy[i] << 16);
        if ((err = hipMalloc(&L.img1, P * plane)) != hipSuccess || (err = hipMalloc(&L.img2, P * plane)) != hipSuccess ||
            (err = hipMalloc(&L.small[0], P * own_blocks * sizeof(mv_t))) != hipSuccess ||
            (err = hipMalloc(&L.small[1], P * own_blocks * sizeof(mv_t))) != hipSuccess ||
            (err = hipMalloc(&L.pred, P * own_blocks * sizeof(mv_t))) != hipSuccess ||
            (err = hipMalloc(&L.fix_list, P * own_blocks * sizeof(uint32_t))) != hipSuccess ||
            (err = hipMalloc(&L.fix_count, P * 64)) != hipSuccess ||
            (err = hipMemset(L.fix_count, 0, P * 64)) != hipSuccess ||
            (err = hipMalloc(&L.big[0], P * cells * sizeof(mv_t))) != hipSuccess ||
            (err = hipMalloc(&L.big[1], P * cells * sizeof(mv_t))) != hipSuccess ||
            (err = hipMalloc(&L.spiral, packed.size() * 4)) != hipSuccess ||
            (err = hipMemset(L.img1, 0, P * plane)) != hipSuccess || (err = hipMemset(L.img2, 0, P * plane)) != hipSuccess ||
            (err = hipMemcpy(L.spiral, packed.data(), packed.size() * 4, hipMemcpyHostToDevice)) != hipSuccess)
            return cleanup_fail(bbme::fail(BBME_ERR_HIP, "allocating level %d: %s", l, hipGetErrorString(err)));
        if (!((L.block == 8 || L.block == 16 || L.block == 32) && L.range <= 63)) {
            // the generic kernel (block 4 / 64, or a range beyond the strip kernel's packed keys): its window may need more LDS
            // than a kernel gets by default
            if (L.lds_bytes > 48 * 1024) {
                if (L.lds_bytes > 160 * 1024)
                    return cleanup_fail(bbme::fail(BBME_ERR_UNSUPPORTED, "level %d: a %dx%d block with range %d needs %zu bytes of LDS", l,
                                                   L.block, L.block, L.range, L.lds_bytes));
                const int bytes = (int)L.lds_bytes;
                switch (L.block) {
                case 2:  err = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_search_generic<2>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); break;
                case 4:  err = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_search_generic<4>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); break;
                case 8:  err = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_search_generic<8>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); break;
                case 16: err = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_search_generic<16>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); break;
                case 32: err = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_search_generic<32>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); break;
                default: err = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_search_generic<64>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); break;
                }
                if (err != hipSuccess)
                    return cleanup_fail(bbme::fail(BBME_ERR_HIP, "level %d: %zu bytes of LDS for the search window: %s", l, L.lds_bytes, hipGetErrorString(err)));
            }
        } else {
            // the strip kernel reads rank rows dy0 .. dy0+S-1 as 4 x u16 per column group
            SearchPlan plan = plan_search(L.range, L.block, L.block == 32 ? 8 : 16);
            L.fast = true;
            L.rank_pitch = sp.rank_pitch;
            L.nrounds = (int)plan.rounds.size();
            L.fast_pitch_dw = plan.pitch_dw;
            // the window (+ for B <= 16 a copy of the block, 16-byte aligned, for the rim rounds of the tight plan)
            L.fast_lds_bytes = (((size_t)(L.block + 2 * L.range) * plan.pitch_dw + 3) & ~(size_t)3) * 4 + (size_t)L.block * L.block;
            // the device copy of a plan's round codes carries, for strip rounds, where the round's rank entries start in
            // lane_ranks (<< 16, in rows of T entries); lane_ranks itself: per strip round and lane the S entries of 4 ranks
            auto upload_plan = [&](const SearchPlan &p, int T, uint32_t **d_tasks, uint32_t **d_rounds, uint2 **d_ranks) -> int {
                std::vector<uint32_t> codes(p.rounds);
                std::vector<uint16_t> ranks;
                uint32_t cum = 0;
                for (size_t rd = 0; rd < p.rounds.size(); ++rd) {
                    if ((p.rounds[rd] >> 8) != 0) continue;
                    const uint32_t S = p.rounds[rd] & 0xffu;
                    if (cum > 0xffffu) return bbme::fail(BBME_ERR_STATE, "search plan of level %d: too many strip rows", l);
                    codes[rd] |= cum << 16;
                    for (int t = 0; t < T; ++t) {
                        const uint32_t task = p.tasks[rd * (size_t)T + t];
                        for (uint32_t d = 0; d < S; ++d)
                            for (int cc = 0; cc < 4; ++cc) {
                                uint16_t r = 0xffffu;                     // idle lane / padding column: masked in the kernel
                                if (task != 0xffffffffu) {
                                    const int dxi = 4 * (int)(task & 0xffu) + cc, dyi = (int)((task >> 8) & 0xffu) + (int)d;
                                    if (dyi > 2 * L.range)
                                        return bbme::fail(BBME_ERR_STATE, "search plan of level %d: a strip leaves the candidate square", l);
                                    if (dxi < sp.rank_pitch) r = sp.rank_of[(size_t)dyi * sp.rank_pitch + dxi];
                                }
                                ranks.push_back(r);
                            }
                    }
                    cum += S;
                }
                if (ranks.empty()) ranks.assign(4, 0xffffu);
                hipError_t e;
                if ((e = hipMalloc(d_tasks, p.tasks.size() * 4)) != hipSuccess ||
                    (e = hipMalloc(d_rounds, codes.size() * 4)) != hipSuccess ||
                    (e = hipMalloc(d_ranks, ranks.size() * 2 + 64)) != hipSuccess ||
                    (e = hipMemcpy(*d_tasks, p.tasks.data(), p.tasks.size() * 4, hipMemcpyHostToDevice)) != hipSuccess ||
                    (e = hipMemcpy(*d_rounds, codes.data(), codes.size() * 4, hipMemcpyHostToDevice)) != hipSuccess ||
                    (e = hipMemcpy(*d_ranks, ranks.data(), ranks.size() * 2, hipMemcpyHostToDevice)) != hipSuccess)
                    return bbme::fail(BBME_ERR_HIP, "allocating search plan of level %d: %s", l, hipGetErrorString(e));
                return BBME_OK;
            };
            if ((err = hipMalloc(&L.rank_of, sp.rank_of.size() * 2 + 64)) != hipSuccess ||
                (err = hipMemcpy(L.rank_of, sp.rank_of.data(), sp.rank_of.size() * 2, hipMemcpyHostToDevice)) != hipSuccess)
                return cleanup_fail(bbme::fail(BBME_ERR_HIP, "allocating search plan of level %d: %s", l, hipGetErrorString(err)));
            if (int rc = upload_plan(plan, 64, &L.tasks, &L.rounds, &L.lane_ranks)) return cleanup_fail(rc);
            // shorter strips, so that the 128 lanes of two waves have a full round of them
            SearchPlan plan2 = plan_search(L.range, L.block, 8, 128);
            L.nrounds2 = (int)plan2.rounds.size();
            // a round of strips of S rows walks S + B - 1 window rows: the split only pays where it shortens a wave's walk
            // (+-32 at B <= 16: 0.6x; +-16: the square is too small to fill 128 lanes with tall strips, 0.94-1.0x -- measured slower)
            auto walk = [&](const SearchPlan &p) {
                int w = 0;
                for (uint32_t code : p.rounds) w += (code >> 8) ? L.block / 4 : (int)(code & 0xffu) + L.block - 1;   // rim rounds are short
                return w;
            };
            L.split_pays = 5 * walk(plan2) <= 4 * walk(plan);
            if (plan2.pitch_dw != plan.pitch_dw) return cleanup_fail(bbme::fail(BBME_ERR_STATE, "search plans disagree on the window pitch"));
            if (int rc = upload_plan(plan2, 128, &L.tasks2, &L.rounds2, &L.lane_ranks2)) return cleanup_fail(rc);
        }
    }
    // pitch = 33 (mod 64) words: consecutive blocks land 132 bytes (mod 256) apart
    c->own_pitch = (uint32_t)(((max_blocks + 31) / 32 + 63) / 64 * 64 + 33);
    const size_t bit_words = (size_t)c->own_pitch * 32;
    c->flag_bytes = (max_blocks + 2047) / 2048 * 2048 + 2048;             // whole 16-flag segments (k_reg_solve), zero beyond the grid
    c->flow_stride = (size_t)g.padded_width * g.padded_height * 2;        // floats
    c->list_stride = (uint32_t)max_blocks; c->own_stride = (uint32_t)bit_words;
    const size_t flow_bytes = P * c->flow_stride * sizeof(float);
    if ((err = hipMalloc(&c->flow, flow_bytes)) != hipSuccess ||
        (err = hipMalloc(&c->list[0], P * max_blocks * 4)) != hipSuccess ||
        (err = hipMalloc(&c->list[1], P * max_blocks * 4)) != hipSuccess ||
        (err = hipMalloc(&c->flags[0], P * c->flag_bytes)) != hipSuccess ||
        (err = hipMalloc(&c->flags[1], P * c->flag_bytes)) != hipSuccess ||
        (err = hipMemset(c->flags[0], 0, P * c->flag_bytes)) != hipSuccess ||
        (err = hipMemset(c->flags[1], 0, P * c->flag_bytes)) != hipSuccess ||
        (err = hipMalloc(&c->own, P * bit_words * 4)) != hipSuccess ||
        (err = hipMalloc(&c->counters, P * 256)) != hipSuccess ||
        (err = hipMemset(c->own, 0, P * bit_words * 4)) != hipSuccess ||
        (err = hipMemset(c->counters, 0, P * 256)) != hipSuccess ||
        (err = hipMemset(c->flow, 0, flow_bytes)) != hipSuccess)
        return cleanup_fail(bbme::fail(BBME_ERR_HIP, "allocating work buffers: %s", hipGetErrorString(err)));
    if (c->use_memo && c->memo_blocks) {
        c->memo_stride = (uint32_t)round64(c->memo_blocks << kMemoSlotShift);
        // every slot starts as "nothing known" (the MV half of the word is what counts: 0x80008000 is never a motion vector)
        if ((err = hipMalloc(&c->memo, P * c->memo_stride * sizeof(unsigned long long))) != hipSuccess ||
            (err = hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(c->memo), (int)kMemoNoMv, P * c->memo_stride * 2)) != hipSuccess)
            return cleanup_fail(bbme::fail(BBME_ERR_HIP, "allocating the SAD memo: %s", hipGetErrorString(err)));
    }
    HIP_TRY(hipDeviceSynchronize());
    bbme::clear_error();
    *out = c;
    return BBME_OK;
}

int bbme_destroy(bbme_ctx *c)
{
    if (!c) return BBME_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    drop_graph(c);
    for (Level &L : c->lv) {
        (void)hipFree(L.img1); (void)hipFree(L.img2);
        (void)hipFree(L.small[0]); (void)hipFree(L.small[1]); (void)hipFree(L.pred);
        (void)hipFree(L.fix_list); (void)hipFree(L.fix_count);
        (void)hipFree(L.big[0]); (void)hipFree(L.big[1]); (void)hipFree(L.spiral);
        (void)hipFree(L.rank_of); (void)hipFree(L.tasks); (void)hipFree(L.rounds); (void)hipFree(L.tasks2); (void)hipFree(L.rounds2);
        (void)hipFree(L.lane_ranks); (void)hipFree(L.lane_ranks2);
    }
    (void)hipFree(c->flow);
    (void)hipFree(c->epe_scratch);
    (void)hipFree(c->raw[0]); (void)hipFree(c->raw[1]);
    (void)hipFree(c->list[0]); (void)hipFree(c->list[1]);
    (void)hipFree(c->own);
    (void)hipFree(c->flags[0]); (void)hipFree(c->flags[1]);
    (void)hipFree(c->counters);
    (void)hipFree(c->memo);
    if (c->side_stream) { (void)hipStreamSynchronize(c->side_stream); (void)hipStreamDestroy(c->side_stream); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return BBME_OK;
}

int bbme_set_stream(bbme_ctx *c, void *hip_stream)
{
    if (int rc = check_ctx(c)) return rc;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_graph(c);
    if (c->own_stream) { (void)hipStreamDestroy(c->stream); c->own_stream = false; }
    c->stream = (hipStream_t)hip_stream;
    if (!c->stream) {
        HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    return BBME_OK;
}

int bbme_set_search_mode(bbme_ctx *c, int mode)
{
    if (int rc = check_ctx(c)) return rc;
    if (mode != BBME_SEARCH_SPIRAL && mode != BBME_SEARCH_RASTER) return bbme::fail(BBME_ERR_INVALID, "search mode %d", mode);
    if (c->raster_search == (mode == BBME_SEARCH_RASTER)) return BBME_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_graph(c);                                   // other kernels in the launch sequence
    c->raster_search = mode == BBME_SEARCH_RASTER;
    return BBME_OK;
}

int bbme_set_regularizer_mode(bbme_ctx *c, int mode)
{
    if (int rc = check_ctx(c)) return rc;
    if (mode != BBME_REG_EXACT && mode != BBME_REG_JACOBI) return bbme::fail(BBME_ERR_INVALID, "regulariser mode %d", mode);
    if (c->jacobi == (mode == BBME_REG_JACOBI)) return BBME_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_graph(c);
    c->jacobi = mode == BBME_REG_JACOBI;
    return BBME_OK;
}

int bbme_set_speculation(bbme_ctx *c, int enabled)
{
    if (int rc = check_ctx(c)) return rc;
    if (c->speculate == (enabled != 0)) return BBME_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_graph(c);                                   // the launch sequence changes
    c->speculate = enabled != 0;
    if (!c->speculate && c->side_stream) { (void)hipStreamDestroy(c->side_stream); c->side_stream = nullptr; }
    return BBME_OK;
}

int bbme_set_relaxation(bbme_ctx *c, int enabled)
{
    if (int rc = check_ctx(c)) return rc;
    if (c->relax == (enabled != 0)) return BBME_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_graph(c);                                   // the launch sequence changes
    c->relax = enabled != 0;
    return BBME_OK;
}

int bbme_wait_for_stream(bbme_ctx *c, void *producer_stream)
{
    if (int rc = check_ctx(c)) return rc;
    HIP_TRY(hipSetDevice(c->device));
    hipEvent_t ev;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, static_cast<hipStream_t>(producer_stream));
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, ev, 0);
    (void)hipEventDestroy(ev);                      // released once the recorded work has completed
    if (e != hipSuccess) return bbme::fail(BBME_ERR_HIP, "bbme_wait_for_stream: %s", hipGetErrorString(e));
    return BBME_OK;
}

int bbme_get_stream(bbme_ctx *c, void **hip_stream)
{
    if (int rc = check_ctx(c)) return rc;
    if (!hip_stream) return bbme::fail(BBME_ERR_INVALID, "null output");
    *hip_stream = c->stream;
    return BBME_OK;
}

int bbme_get_geometry(const bbme_ctx *c, int *pw, int *ph, int *px, int *py)
{
    if (int rc = check_ctx(c)) return rc;
    if (pw) *pw = c->geom.padded_width;
    if (ph) *ph = c->geom.padded_height;
    if (px) *px = c->geom.pad_x;
    if (py) *py = c->geom.pad_y;
    return BBME_OK;
}

int bbme_level_geometry(const bbme_ctx *c, int level, int *w, int *h, int *b, int *s)
{
    if (int rc = check_level(c, level)) return rc;
    const Level &L = c->lv[level];
    if (w) *w = L.width;
    if (h) *h = L.height;
    if (b) *b = L.block;
    if (s) *s = L.search;
    return BBME_OK;
}

int bbme_set_frames_host(bbme_ctx *c, const uint8_t *image1, const uint8_t *image2, int pitch)
{
    if (int rc = bbme_set_frames_host_async(c, 0, image1, image2, pitch)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));        // the caller may re-use its buffers
    return BBME_OK;
}

int bbme_set_frames_host_pair(bbme_ctx *c, int pair, const uint8_t *image1, const uint8_t *image2, int pitch)
{
    if (int rc = bbme_set_frames_host_async(c, pair, image1, image2, pitch)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BBME_OK;
}

int bbme_set_frames_host_async(bbme_ctx *c, int pair, const uint8_t *image1, const uint8_t *image2, int pitch)
{
    if (int rc = check_ctx(c)) return rc;
    if (!image1 || !image2 || pitch < c->geom.width || pair < 0 || pair >= c->batch)
        return bbme::fail(BBME_ERR_INVALID, "bbme_set_frames_host: bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    // the frames as they are go to HBM; zero border and pyrDown cascade run there (bbme_set_frames_device) -- the same
    // integers as bbme_pad_zero_host / bbme_pyr_down_host produce, without 18 ms of single-threaded host filtering at 4K
    const Geometry &g = c->geom;
    const size_t bytes = (size_t)g.width * g.height;
    c->raw_stride = (bytes + 64 + 255) / 256 * 256;
    const uint8_t *src[2] = {image1, image2};
    for (int i = 0; i < 2; ++i) {
        if (!c->raw[i]) HIP_TRY(hipMalloc(&c->raw[i], c->raw_stride * c->batch));
        HIP_TRY(hipMemcpy2DAsync(c->raw[i] + pair * c->raw_stride, g.width, src[i], pitch, g.width, g.height, hipMemcpyHostToDevice, c->stream));
    }
    // no host wait: with pinned source buffers the upload, the border, the pyramid and an estimate behind them overlap whatever
    // the host does next; the buffers must stay untouched until the context's stream has passed this point
    return bbme_set_frames_device_pair(c, pair, c->raw[0] + pair * c->raw_stride, c->raw[1] + pair * c->raw_stride, g.width);
}

int bbme_set_frames_device(bbme_ctx *c, const uint8_t *d_image1, const uint8_t *d_image2, int pitch)
{
    return bbme_set_frames_device_pair(c, 0, d_image1, d_image2, pitch);
}

int bbme_set_frames_device_pair(bbme_ctx *c, int pair, const uint8_t *d_image1, const uint8_t *d_image2, int pitch)
{
    if (int rc = check_ctx(c)) return rc;
    if (!d_image1 || !d_image2 || pitch < c->geom.width || pair < 0 || pair >= c->batch)
        return bbme::fail(BBME_ERR_INVALID, "bbme_set_frames_device: bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    const Geometry &g = c->geom;
    const size_t pp_ = (size_t)pair;
    // both frames per launch: zero border into the level-0 planes, then the pyrDown cascade
    Level &L0 = c->lv[0];
    PlanePair pp{{d_image1, d_image2}, {L0.img1 + pp_ * L0.plane_stride, L0.img2 + pp_ * L0.plane_stride}};
    const long long chunks = (long long)((L0.width + 15) / 16) * L0.height;
    hipLaunchKernelGGL(k_pad_zero, dim3((unsigned)((chunks + 255) / 256), 2), dim3(256), 0, c->stream,
                       pp, g.width, g.height, pitch, g.pad_x, g.pad_y, L0.width, L0.height);
    for (size_t l = 1; l < c->lv.size(); ++l) {
        Level &P = c->lv[l - 1], &L = c->lv[l];
        PlanePair q{{P.img1 + pp_ * P.plane_stride, P.img2 + pp_ * P.plane_stride}, {L.img1 + pp_ * L.plane_stride, L.img2 + pp_ * L.plane_stride}};
        if (P.width % 8 == 0) {
            const long long n = (long long)(L.width / 4) * L.height;
            hipLaunchKernelGGL(k_pyr_down4, dim3((unsigned)((n + 255) / 256), 2), dim3(256), 0, c->stream, q, P.width, P.height);
        } else {
            const long long n = (long long)L.width * L.height;
            hipLaunchKernelGGL(k_pyr_down, dim3((unsigned)((n + 255) / 256), 2), dim3(256), 0, c->stream, q, P.width, P.height);
        }
    }
    HIP_TRY(hipGetLastError());
    c->frames_mask |= 1ull << pair;
    c->memo_block = 0;                                  // new planes: what the SAD memo holds is no longer true
    return BBME_OK;
}

int bbme_level_planes_device(bbme_ctx *c, int level, uint8_t **d1, uint8_t **d2)
{
    if (int rc = single_pair_only(c, "bbme_level_planes_device")) return rc;
    if (int rc = check_level(c, level)) return rc;
    if (d1) *d1 = c->lv[level].img1;
    if (d2) *d2 = c->lv[level].img2;
    c->frames_mask |= 1ull;        // the caller fills them in place (pair 0)
    c->memo_block = 0;
    return BBME_OK;
}

int bbme_set_level_planes_host(bbme_ctx *c, int level, const uint8_t *image1, const uint8_t *image2)
{
    if (int rc = single_pair_only(c, "bbme_set_level_planes_host")) return rc;
    if (int rc = check_level(c, level)) return rc;
    if (!image1 || !image2) return bbme::fail(BBME_ERR_INVALID, "null plane");
    HIP_TRY(hipSetDevice(c->device));
    Level &L = c->lv[level];
    HIP_TRY(hipMemcpyAsync(L.img1, image1, (size_t)L.width * L.height, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(L.img2, image2, (size_t)L.width * L.height, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->frames_mask |= 1ull;
    c->memo_block = 0;
    return BBME_OK;
}

int bbme_get_level_planes_host(bbme_ctx *c, int level, uint8_t *image1, uint8_t *image2)
{
    if (int rc = single_pair_only(c, "bbme_get_level_planes_host")) return rc;
    if (int rc = check_level(c, level)) return rc;
    HIP_TRY(hipSetDevice(c->device));
    Level &L = c->lv[level];
    if (image1) HIP_TRY(hipMemcpyAsync(image1, L.img1, (size_t)L.width * L.height, hipMemcpyDeviceToHost, c->stream));
    if (image2) HIP_TRY(hipMemcpyAsync(image2, L.img2, (size_t)L.width * L.height, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BBME_OK;
}

int bbme_estimate(bbme_ctx *c)
{
    if (int rc = check_ctx(c)) return rc;
    if (!c->frames_set()) return bbme::fail(BBME_ERR_STATE, "bbme_estimate: no frames set (every pair of a batch needs its frames)");
    HIP_TRY(hipSetDevice(c->device));
    if (c->profiling) { const int rc = profiled_pyramid(c); c->memo_block = 0; return rc; }
    if (!c->use_graph) { const int rc = enqueue_pyramid(c, c->speculate); c->memo_block = 0; return rc; }
    if (!c->graph_exec) {
        // the launch sequence is fixed (no host decisions inside), so capture it once
        hipGraph_t graph = nullptr;
        HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        int rc = enqueue_pyramid(c, c->speculate);
        hipError_t e = hipStreamEndCapture(c->stream, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess) return bbme::fail(BBME_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
        e = hipGraphInstantiate(&c->graph_exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { c->graph_exec = nullptr; return bbme::fail(BBME_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
    } else {
        // keep the host-side grid bookkeeping in step with what the graph replays
        for (Level &L : c->lv) { L.cur_grid = L.final_grid(); L.cur_block = 2; }
    }
    HIP_TRY(hipGraphLaunch(c->graph_exec, c->stream));
    // after a pyramid the memo describes level 0 at its last memoised block size; a later stage call starts afresh
    c->memo_block = 0;
    return BBME_OK;
}

int bbme_synchronize(bbme_ctx *c)
{
    if (int rc = check_ctx(c)) return rc;
    HIP_TRY(hipSetDevice(c->device));
    return check_converged(c);
}

int bbme_batch_size(const bbme_ctx *c, int *pairs)
{
    if (int rc = check_ctx(c)) return rc;
    if (!pairs) return bbme::fail(BBME_ERR_INVALID, "null output");
    *pairs = c->batch;
    return BBME_OK;
}

static int check_pair(const bbme_ctx *c, int pair)
{
    if (int rc = check_ctx(c)) return rc;
    if (pair < 0 || pair >= c->batch) return bbme::fail(BBME_ERR_INVALID, "pair %d of a batch of %d", pair, c->batch);
    return BBME_OK;
}

int bbme_flow_device(bbme_ctx *c, const float **d_flow) { return bbme_flow_device_pair(c, 0, d_flow); }

int bbme_flow_device_pair(bbme_ctx *c, int pair, const float **d_flow)
{
    if (int rc = check_pair(c, pair)) return rc;
    if (!d_flow) return bbme::fail(BBME_ERR_INVALID, "null output");
    *d_flow = c->flow + (size_t)pair * c->flow_stride;
    return BBME_OK;
}

int bbme_get_flow_host(bbme_ctx *c, float *flow) { return bbme_get_flow_host_pair(c, 0, flow); }

int bbme_get_flow_host_pair(bbme_ctx *c, int pair, float *flow)
{
    if (int rc = check_pair(c, pair)) return rc;
    if (!flow) return bbme::fail(BBME_ERR_INVALID, "null output");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(flow, c->flow + (size_t)pair * c->flow_stride, c->flow_stride * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    return check_converged(c);
}

int bbme_get_cells_host(bbme_ctx *c, int16_t *cells) { return bbme_get_cells_host_pair(c, 0, cells); }

int bbme_get_cells_host_pair(bbme_ctx *c, int pair, int16_t *cells)
{
    if (int rc = check_pair(c, pair)) return rc;
    if (!cells) return bbme::fail(BBME_ERR_INVALID, "null output");
    Level &L = c->lv[0];
    if (L.cur_block != 2) return bbme::fail(BBME_ERR_STATE, "level 0 is not at 2x2 cells");
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = (size_t)(L.width / 2) * (L.height / 2);
    HIP_TRY(hipMemcpyAsync(cells, L.cur_grid + (size_t)pair * L.grid_stride(L.cur_grid), n * sizeof(mv_t), hipMemcpyDeviceToHost, c->stream));
    return check_converged(c);
}

int bbme_cells_device(bbme_ctx *c, const int16_t **d_cells) { return bbme_cells_device_pair(c, 0, d_cells); }

int bbme_cells_device_pair(bbme_ctx *c, int pair, const int16_t **d_cells)
{
    if (int rc = check_pair(c, pair)) return rc;
    if (!d_cells) return bbme::fail(BBME_ERR_INVALID, "null output");
    // two sweeps per block size always leave the final field of a level in the same buffer (Level::final_grid)
    const Level &L = c->lv[0];
    *d_cells = reinterpret_cast<const int16_t *>(L.final_grid() + (size_t)pair * L.grid_stride(L.final_grid()));
    return BBME_OK;
}

int bbme_expand_cells_device(bbme_ctx *c, const int16_t *d_cells, float *d_flow)
{
    return bbme_expand_cells_device_on(c, d_cells, d_flow, nullptr);
}

int bbme_expand_cells_device_on(bbme_ctx *c, const int16_t *d_cells, float *d_flow, void *hip_stream)
{
    if (int rc = check_ctx(c)) return rc;
    if (!d_cells || !d_flow) return bbme::fail(BBME_ERR_INVALID, "null pointer");
    HIP_TRY(hipSetDevice(c->device));
    Level &L = c->lv[0];
    const int cc = L.width / 2, cr = L.height / 2;
    const long long threads = (long long)cc * cr * 2;
    hipLaunchKernelGGL(k_expand, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream,
                       reinterpret_cast<const mv_t *>(d_cells), cc, cr, d_flow, L.width, 0u, (size_t)0);
    HIP_TRY(hipGetLastError());
    return BBME_OK;
}

int bbme_calculate_mse_device(bbme_ctx *c, const float *d_gtruth, int gt_width, int gt_height, int scale, double *out)
{
    if (int rc = single_pair_only(c, "bbme_calculate_mse_device")) return rc;
    if (int rc = check_ctx(c)) return rc;
    if (!d_gtruth || !out || gt_width < 1 || gt_height < 1 || scale < 1)
        return bbme::fail(BBME_ERR_INVALID, "bbme_calculate_mse_device: bad arguments");
    Level &L = c->lv[0];
    if (L.cur_block != 2) return bbme::fail(BBME_ERR_STATE, "level 0 has not been regularised down to 2x2 blocks");
    if ((long long)(gt_width - 1) * scale >= c->geom.width || (long long)(gt_height - 1) * scale >= c->geom.height)
        return bbme::fail(BBME_ERR_INVALID, "ground truth %dx%d at scale %d does not fit the %dx%d frame",
                          gt_width, gt_height, scale, c->geom.width, c->geom.height);
    HIP_TRY(hipSetDevice(c->device));
    constexpr int kMaxGroups = 512;
    const long long n = (long long)gt_width * gt_height;
    const int groups = (int)std::min<long long>(kMaxGroups, (n + 255) / 256);
    if (!c->epe_scratch) HIP_TRY(hipMalloc(&c->epe_scratch, kMaxGroups * (sizeof(double) + sizeof(unsigned long long))));
    double *d_sum = c->epe_scratch;
    unsigned long long *d_cnt = reinterpret_cast<unsigned long long *>(d_sum + kMaxGroups);
    hipLaunchKernelGGL(k_epe, dim3(groups), dim3(256), 0, c->stream, L.cur_grid, L.width / 2,
                       c->geom.pad_x, c->geom.pad_y, scale, d_gtruth, gt_width, gt_height, d_sum, d_cnt);
    std::vector<double> h_sum(kMaxGroups);
    std::vector<unsigned long long> h_cnt(kMaxGroups);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) err = hipMemcpyAsync(h_sum.data(), d_sum, groups * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (err == hipSuccess) err = hipMemcpyAsync(h_cnt.data(), d_cnt, groups * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
    if (err != hipSuccess) return bbme::fail(BBME_ERR_HIP, "bbme_calculate_mse_device: %s", hipGetErrorString(err));
    if (int rc = check_converged(c)) return rc;
    double error = 0;
    unsigned long long count = 0;
    for (int i = 0; i < groups; ++i) { error += h_sum[i]; count += h_cnt[i]; }
    *out = error / (double)count;                     // 0/0 = NaN when no pixel is known, as in the reference (:330)
    return BBME_OK;
}

int bbme_stage_search(bbme_ctx *c, int level)
{
    if (int rc = single_pair_only(c, "bbme_stage_search")) return rc;
    if (int rc = check_level(c, level)) return rc;
    if (!c->frames_set()) return bbme::fail(BBME_ERR_STATE, "no frames set");
    HIP_TRY(hipSetDevice(c->device));
    return launch_search(c, level);
}

int bbme_stage_regularize(bbme_ctx *c, int level, int block, int mult)
{
    if (int rc = single_pair_only(c, "bbme_stage_regularize")) return rc;
    if (int rc = check_level(c, level)) return rc;
    if (!c->frames_set()) return bbme::fail(BBME_ERR_STATE, "no frames set");
    HIP_TRY(hipSetDevice(c->device));
    return launch_sweep(c, level, block, mult, true);       // with the solver's counters (bbme_sweep_stats)
}

int bbme_stage_get_mvs(bbme_ctx *c, int level, int block, int16_t *mvs)
{
    if (int rc = single_pair_only(c, "bbme_stage_get_mvs")) return rc;
    if (int rc = check_level(c, level)) return rc;
    Level &L = c->lv[level];
    if (!mvs) return bbme::fail(BBME_ERR_INVALID, "null output");
    if (L.cur_block == 0) return bbme::fail(BBME_ERR_STATE, "level %d has no MV grid yet", level);
    if (block < 1 || block > L.cur_block || (L.cur_block % block))
        return bbme::fail(BBME_ERR_INVALID, "block %d does not divide the grid's block size %d", block, L.cur_block);
    HIP_TRY(hipSetDevice(c->device));
    const int rows = L.height / L.cur_block, cols = L.width / L.cur_block;
    std::vector<mv_t> host((size_t)rows * cols);
    HIP_TRY(hipMemcpyAsync(host.data(), L.cur_grid, host.size() * sizeof(mv_t), hipMemcpyDeviceToHost, c->stream));
    if (int rc = check_converged(c)) return rc;
    // divide_blocks (:845-862) repeated: every finer block inherits its parent's MV
    const int f = L.cur_block / block, orows = rows * f, ocols = cols * f;
    for (int r = 0; r < orows; ++r)
        for (int q = 0; q < ocols; ++q) {
            const mv_t m = host[(size_t)(r / f) * cols + q / f];
            mvs[2 * ((size_t)r * ocols + q)] = (int16_t)(m & 0xffffu);
            mvs[2 * ((size_t)r * ocols + q) + 1] = (int16_t)(m >> 16);
        }
    return BBME_OK;
}

int bbme_stage_set_mvs(bbme_ctx *c, int level, int block, const int16_t *mvs)
{
    if (int rc = single_pair_only(c, "bbme_stage_set_mvs")) return rc;
    if (int rc = check_level(c, level)) return rc;
    Level &L = c->lv[level];
    if (!mvs) return bbme::fail(BBME_ERR_INVALID, "null input");
    if (block < 2 || block > L.block || (block & (block - 1)))
        return bbme::fail(BBME_ERR_INVALID, "block %d is not a power of two in 2..%d", block, L.block);
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = (size_t)(L.height / block) * (L.width / block);
    std::vector<mv_t> host(n);
    for (size_t i = 0; i < n; ++i) host[i] = ((uint32_t)(uint16_t)mvs[2 * i]) | ((uint32_t)(uint16_t)mvs[2 * i + 1] << 16);
    L.cur_grid = block == L.block ? L.small[0] : L.big[0];
    L.cur_block = block;
    HIP_TRY(hipMemcpyAsync(L.cur_grid, host.data(), n * sizeof(mv_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BBME_OK;
}

int bbme_stage_expand(bbme_ctx *c)
{
    if (int rc = single_pair_only(c, "bbme_stage_expand")) return rc;
    if (int rc = check_ctx(c)) return rc;
    HIP_TRY(hipSetDevice(c->device));
    return launch_expand(c);
}

int bbme_last_sweep_passes(bbme_ctx *c, int *passes)
{
    if (int rc = single_pair_only(c, "bbme_last_sweep_passes")) return rc;
    if (int rc = check_ctx(c)) return rc;
    if (!passes) return bbme::fail(BBME_ERR_INVALID, "null output");
    HIP_TRY(hipSetDevice(c->device));
    uint32_t host[8];
    HIP_TRY(hipMemcpyAsync(host, c->counters, sizeof host, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    passes[0] = (int)host[3];
    passes[1] = (int)host[4];
    if (host[5]) return bbme::fail(BBME_ERR_STATE, "a regulariser sweep hit its pass cap without converging");
    return BBME_OK;
}

int bbme_sweep_stats(bbme_ctx *c, unsigned *stats)
{
    if (int rc = single_pair_only(c, "bbme_sweep_stats")) return rc;
    if (int rc = check_ctx(c)) return rc;
    if (!stats) return bbme::fail(BBME_ERR_INVALID, "null output");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(stats, c->counters, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BBME_OK;
}

int bbme_set_profiling(bbme_ctx *c, int enabled)
{
    if (int rc = check_ctx(c)) return rc;
    c->profiling = enabled != 0;
    return BBME_OK;
}

int bbme_get_timings(bbme_ctx *c, float *total, float *search, float *reg, float *expand, float *search0)
{
    if (int rc = check_ctx(c)) return rc;
    if (total) *total = c->t_total;
    if (search) *search = c->t_search;
    if (reg) *reg = c->t_reg;
    if (expand) *expand = c->t_expand;
    if (search0) *search0 = c->t_search0;
    return BBME_OK;
}

int bbme_probe_rates(int device, double *gops)
{
    if (!gops) return bbme::fail(BBME_ERR_INVALID, "null output");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return bbme::fail(BBME_ERR_HIP, "no HIP device %d", device);
    HIP_TRY(hipSetDevice(device));
    uint32_t *out = nullptr;
    HIP_TRY(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    const int iters = 4096, grid = 256 * 8;          // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    for (int which = 0; which < 4; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            HIP_TRY(hipEventRecord(e0, 0));
            if (which == 0) hipLaunchKernelGGL(k_probe_rate<0>, dim3(grid), dim3(256), 0, 0, out, iters, 7u + rep);
            else if (which == 1) hipLaunchKernelGGL(k_probe_rate<1>, dim3(grid), dim3(256), 0, 0, out, iters, 7u + rep);
            else if (which == 2) hipLaunchKernelGGL(k_probe_rate<2>, dim3(grid), dim3(256), 0, 0, out, iters, 7u + rep);
            else hipLaunchKernelGGL(k_probe_rate<5>, dim3(grid), dim3(256), 0, 0, out, iters, 7u + rep);
            HIP_TRY(hipEventRecord(e1, 0));
            HIP_TRY(hipEventSynchronize(e1));
        }
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        // wave-instructions per second over the whole chip, in units of 1e9 (mixed runs: QSADs only)
        gops[which] = (double)grid * 4 * iters * 8 / (ms * 1e-3) / 1e9;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(out);
    return BBME_OK;
}

int bbme_probe_search_loops(int device, double *tabs2)
{
    if (!tabs2) return bbme::fail(BBME_ERR_INVALID, "null output");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return bbme::fail(BBME_ERR_HIP, "no HIP device %d", device);
    HIP_TRY(hipSetDevice(device));
    uint32_t *out = nullptr;
    HIP_TRY(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    const int passes = 128, grid = 32768;
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            HIP_TRY(hipEventRecord(e0, 0));
            if (which == 0) hipLaunchKernelGGL(k_probe_search_loop<false>, dim3(grid), dim3(64), 6912 + 21 * 32 * 4, 0, out, passes, 3u + rep);
            else hipLaunchKernelGGL(k_probe_search_loop<true>, dim3(grid), dim3(64), 27648 + 16 * 5 * 32 + 16 * 5 * 31, 0, out, passes, 3u + rep);
            HIP_TRY(hipEventRecord(e1, 0));
            HIP_TRY(hipEventSynchronize(e1));
        }
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        // abs-diffs: per lane and pass 16 x 16 x 4 instructions of 16 (QSAD: four dx at once) or 4 (v_sad_u8) abs-diffs
        const double absdiff = (double)grid * 64 * passes * 1024 * (which == 0 ? 16 : 4);
        tabs2[which] = absdiff / (ms * 1e-3) / 1e12;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(out);
    return BBME_OK;
}

int bbme_probe_latency(int device, unsigned long long *out9)
{
    if (!out9) return bbme::fail(BBME_ERR_INVALID, "null output");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return bbme::fail(BBME_ERR_HIP, "no HIP device %d", device);
    HIP_TRY(hipSetDevice(device));
    const uint32_t nwords = 1u << 18;                       // 1 MiB: beyond L1, inside an L2
    std::vector<uint32_t> host(nwords);
    uint32_t x = 12345;
    for (uint32_t i = 0; i < nwords; ++i) { x = x * 1664525u + 1013904223u; host[i] = x; }
    uint32_t *buf = nullptr; unsigned long long *out = nullptr;
    HIP_TRY(hipMalloc(&buf, nwords * 4)); HIP_TRY(hipMalloc(&out, 9 * 8));
    HIP_TRY(hipMemcpy(buf, host.data(), nwords * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_probe_latency, dim3(1), dim3(64), 0, 0, buf, nwords, out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out9, out, 9 * 8, hipMemcpyDeviceToHost));
    (void)hipFree(buf); (void)hipFree(out);
    return BBME_OK;
}

int bbme_probe_xcd(int device, int *xcds_seen, int *violations)
{
    if (!xcds_seen || !violations) return bbme::fail(BBME_ERR_INVALID, "null output");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return bbme::fail(BBME_ERR_HIP, "no HIP device %d", device);
    HIP_TRY(hipSetDevice(device));
    const int n = 4096;
    std::vector<uint32_t> host(n);
    uint32_t *d = nullptr;
    HIP_TRY(hipMalloc(&d, n * 4));
    hipLaunchKernelGGL(k_probe_xcc, dim3(n), dim3(64), 0, 0, d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(host.data(), d, n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    uint32_t seen = 0;
    *violations = 0;
    for (int b = 0; b < n; ++b) {
        seen |= 1u << host[b];
        if (host[b] != host[b & 7]) ++*violations;
    }
    *xcds_seen = __builtin_popcount(seen);
    return BBME_OK;
}

int bbme_calibrate_read(int device, unsigned mbytes, int repeats)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return bbme::fail(BBME_ERR_HIP, "no HIP device %d", device);
    if (mbytes == 0 || mbytes > 16384 || repeats < 1) return bbme::fail(BBME_ERR_INVALID, "bbme_calibrate_read: bad size");
    HIP_TRY(hipSetDevice(device));
    const size_t bytes = (size_t)mbytes << 20, n = bytes / 4;
    uint32_t *buf = nullptr, *out = nullptr;
    HIP_TRY(hipMalloc(&buf, bytes));
    HIP_TRY(hipMalloc(&out, 64));
    HIP_TRY(hipMemset(buf, 1, bytes));
    HIP_TRY(hipDeviceSynchronize());
    for (int i = 0; i < repeats; ++i)
        hipLaunchKernelGGL(k_calib_read_dword, dim3(256 * 8), dim3(256), 0, 0, buf, n, out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(buf); (void)hipFree(out);
    return BBME_OK;
}

int bbme_selftest_isa(int device, int *mismatches)
{
    if (!mismatches) return bbme::fail(BBME_ERR_INVALID, "null output");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return bbme::fail(BBME_ERR_HIP, "no HIP device %d", device);
    HIP_TRY(hipSetDevice(device));
    const int n = 1 << 16;
    std::vector<uint32_t> a(n), b(n), cc(n), sad(n), al(n), s16(n);
    std::vector<unsigned long long> qs(n);
    uint32_t x = 0x12345678u;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
    for (int i = 0; i < n; ++i) { a[i] = rnd(); b[i] = rnd(); cc[i] = rnd(); }
    uint32_t *da, *db, *dc, *dsad, *dal, *ds16; unsigned long long *dqs;
    HIP_TRY(hipMalloc(&da, n * 4)); HIP_TRY(hipMalloc(&db, n * 4)); HIP_TRY(hipMalloc(&dc, n * 4));
    HIP_TRY(hipMalloc(&dsad, n * 4)); HIP_TRY(hipMalloc(&dal, n * 4)); HIP_TRY(hipMalloc(&ds16, n * 4));
    HIP_TRY(hipMalloc(&dqs, n * 8));
    HIP_TRY(hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dc, cc.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe_sad, dim3(n / 256), dim3(256), 0, 0, da, db, dc, dsad, dqs, dal, ds16, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(sad.data(), dsad, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(al.data(), dal, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(s16.data(), ds16, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(qs.data(), dqs, n * 8, hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc); (void)hipFree(dsad); (void)hipFree(dal);
    (void)hipFree(ds16); (void)hipFree(dqs);
    auto absd = [](int p, int q) { return p > q ? p - q : q - p; };
    mismatches[0] = mismatches[1] = mismatches[2] = mismatches[3] = mismatches[4] = 0;
    {   // unaligned dword / x2 / x4 global loads (score_block relies on them)
        const int m = 4096;
        std::vector<uint8_t> bytes(5 * m + 64);
        for (size_t i = 0; i < bytes.size(); ++i) bytes[i] = (uint8_t)rnd();
        std::vector<uint32_t> got(7 * m);
        uint8_t *dp; uint32_t *dout;
        HIP_TRY(hipMalloc(&dp, bytes.size())); HIP_TRY(hipMalloc(&dout, got.size() * 4));
        HIP_TRY(hipMemcpy(dp, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_probe_unaligned, dim3(m / 256), dim3(256), 0, 0, dp, dout, m);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
        (void)hipFree(dp); (void)hipFree(dout);
        auto rd = [&](size_t o) { return (uint32_t)bytes[o] | ((uint32_t)bytes[o + 1] << 8) | ((uint32_t)bytes[o + 2] << 16) | ((uint32_t)bytes[o + 3] << 24); };
        for (int i = 0; i < m; ++i) {
            const size_t o = 5 * (size_t)i + (i & 3);
            bool ok = got[7 * i] == rd(o) && got[7 * i + 1] == rd(o + 1) && got[7 * i + 2] == rd(o + 5);
            for (int k = 0; k < 4; ++k) ok = ok && got[7 * i + 3 + k] == rd(o + 2 + 4 * k);
            if (!ok) ++mismatches[4];
        }
    }
    for (int i = 0; i < n; ++i) {
        uint32_t e = cc[i];
        for (int k = 0; k < 4; ++k) e += absd((a[i] >> (8 * k)) & 255, (b[i] >> (8 * k)) & 255);
        if (e != sad[i]) ++mismatches[0];
        const unsigned long long w = ((unsigned long long)b[i] << 32) | a[i];
        if ((uint32_t)(w >> (8 * (cc[i] & 3))) != al[i]) ++mismatches[1];
        // v_qsad_pk_u16_u8: four SADs of src1's 4 bytes against src0 shifted by 0..3 bytes,
        // each added to the matching 16-bit lane of the accumulator
        const uint32_t ref = cc[i] ^ a[i];
        const unsigned long long acc = ((unsigned long long)(cc[i] & 0x00ff00ffu) << 32) | (cc[i] & 0x0f0f0f0fu);
        unsigned long long eq = 0;
        for (int k = 0; k < 4; ++k) {
            const uint32_t win = (uint32_t)(w >> (8 * k));
            uint32_t s = (uint32_t)((acc >> (16 * k)) & 0xffff);
            for (int q = 0; q < 4; ++q) s += absd((win >> (8 * q)) & 255, (ref >> (8 * q)) & 255);
            eq |= (unsigned long long)(s & 0xffff) << (16 * k);
        }
        if (eq != qs[i]) ++mismatches[2];
        const uint32_t e16 = (cc[i] & 0xffffu) + absd(a[i] & 0xffff, b[i] & 0xffff) + absd(a[i] >> 16, b[i] >> 16);
        if (e16 != s16[i]) ++mismatches[3];
    }
    return BBME_OK;
}

}  // extern "C"
