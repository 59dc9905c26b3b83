// bbme_kernels.hpp -- HIP kernels for gfx950 (CDNA4, wave64).  Included by bbme_device.hip only.
//
// Data layout in HBM (per pyramid level l, all owned by the context):
//   image1/image2 : uint8 planes, pitch == level width W_l (a multiple of 4).
//   MV grids      : one uint32 per block, (dx & 0xffff) | (dy << 16), int16 halves, row-major
//                   (H_l/b) x (W_l/b) for the block size b currently being regularised.  Two
//                   buffers per level, ping-ponged between "old" (read-only in a sweep) and
//                   "est" (being solved).  The reference instead keeps a dense CV_32FC2 field
//                   and a CV_32SC4 SAD cache per pixel (pyramid_level.h:10, motion_framework.h:46).
//   dense flow    : float2 per pixel of level 0 (the cv::Mat calcMotionBlockMatching returns).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bbme {

typedef uint32_t mv_t;   // packed int16 (dx, dy)

__device__ __forceinline__ int mv_x(mv_t m) { return (int)(int16_t)(m & 0xffffu); }
__device__ __forceinline__ int mv_y(mv_t m) { return (int)(int16_t)(m >> 16); }
__device__ __forceinline__ mv_t mv_pack(int x, int y) { return ((uint32_t)x & 0xffffu) | ((uint32_t)y << 16); }

// =======================================================================================
// K1 (generic form): MF::copyMVs + MF::calcLevelBM + MF::find_min_block_spiral
// (motion_framework.cpp:828-843, 226-244, 296-422).  One wavefront per macroblock.
// The (B+2R)^2 search window of image2 and the BxB block of image1 are staged in LDS; each
// lane walks the spiral ranks lane, lane+64, ... (so inside a lane the first strict minimum
// is the lowest rank), then the wave reduces (SAD, rank) lexicographically.
// =======================================================================================
struct SearchArgs {
    const uint8_t *image1, *image2;
    int width, height;          // level size
    int range;                  // R
    int ncand;                  // (2R+1)^2
    const uint32_t *spiral;     // rank -> (dx & 0xffff) | (dy << 16)
    const mv_t *coarse;         // final 2x2-cell grid of level l+1, or nullptr (coarsest level)
    int coarse_cols;            // its row length = W_{l+1} / 2
    int coarse_block;           // B_{l+1}
    mv_t *out;                  // (H/B) x (W/B)
    int cols;                   // W / B
    int pitch_dw;               // LDS window pitch in dwords
};

template <int B>
__global__ __launch_bounds__(64) void k_search_generic(SearchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    constexpr int BW = B / 4;
    const int lane = threadIdx.x;
    const int bc = blockIdx.x % a.cols, br = blockIdx.x / a.cols;
    const int i = br * B, j = bc * B;                     // block origin (row, col)

    // copyMVs: the coarse block covering pixel (i, j) of this level, MV doubled (:836-840)
    int u = 0, v = 0;
    if (a.coarse) {
        const int ci = (i / (2 * a.coarse_block)) * a.coarse_block;
        const int cj = (j / (2 * a.coarse_block)) * a.coarse_block;
        const mv_t m = a.coarse[(size_t)(ci >> 1) * a.coarse_cols + (cj >> 1)];
        u = 2 * mv_x(m); v = 2 * mv_y(m);
    }
    const int px = j + u, py = i + v;                      // :233-234
    mv_t *dst = a.out + (size_t)br * a.cols + bc;
    if (px < 0 || py < 0 || px + B > a.width || py + B > a.height) {   // :304-310 -> zero MV
        if (lane == 0) *dst = 0;
        return;
    }
    const int R = a.range;
    const int wrows = B + 2 * R;
    const int wx0 = px - R, wy0 = py - R;
    const int ax0 = wx0 & ~3;                              // dword-aligned left edge
    const int sh0 = wx0 - ax0;
    uint32_t *win = smem;
    uint32_t *cur = smem + wrows * a.pitch_dw;

    for (int idx = lane; idx < wrows * a.pitch_dw; idx += 64) {
        const int row = idx / a.pitch_dw, k = idx - row * a.pitch_dw;
        const int y = wy0 + row, x = ax0 + 4 * k;
        uint32_t w = 0;
        if (y >= 0 && y < a.height && x >= 0 && x + 4 <= a.width)
            w = *reinterpret_cast<const uint32_t *>(a.image2 + (size_t)y * a.width + x);
        win[idx] = w;
    }
    for (int idx = lane; idx < B * BW; idx += 64) {
        const int row = idx / BW, k = idx - row * BW;
        cur[idx] = *reinterpret_cast<const uint32_t *>(a.image1 + (size_t)(i + row) * a.width + j + 4 * k);
    }
    __syncthreads();

    uint32_t best_sad = 0xffffffffu, best_rank = 0xffffffffu;
    for (int rank = lane; rank < a.ncand; rank += 64) {
        const uint32_t s = a.spiral[rank];
        const int dx = (int)(int16_t)(s & 0xffffu), dy = (int)(int16_t)(s >> 16);
        const int cx = px + dx, cy = py + dy;
        if (cx < 0 || cy < 0 || cx + B > a.width || cy + B > a.height) continue;   // :335 skipped
        const int ox = dx + R + sh0;                       // byte offset inside an LDS row
        const int k0 = ox >> 2, sh = ox & 3;
        const uint32_t *wrow = win + (dy + R) * a.pitch_dw + k0;
        uint32_t sad = 0;
#pragma unroll 2
        for (int r = 0; r < B; ++r) {
            uint32_t lo = wrow[0];
#pragma unroll
            for (int q = 0; q < BW; ++q) {
                const uint32_t hi = wrow[q + 1];
                const uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, sh);
                sad = __builtin_amdgcn_sad_u8(cur[r * BW + q], w, sad);
                lo = hi;
            }
            wrow += a.pitch_dw;
        }
        if (sad < best_sad) { best_sad = sad; best_rank = (uint32_t)rank; }         // strict :339
    }
    // wave reduction of (sad, rank), lexicographic
    unsigned long long key = ((unsigned long long)best_sad << 32) | best_rank;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o);
        key = other < key ? other : key;
    }
    if (lane == 0) {
        const uint32_t s = a.spiral[(uint32_t)key];
        *dst = mv_pack(u + (int)(int16_t)(s & 0xffffu), v + (int)(int16_t)(s >> 16));   // :238-239
    }
}

// =======================================================================================
// K2: MF::regularize_MVs / find_min_candidate / calculate_smoothness / min_energy_candidate
// (motion_framework.cpp:424-662), solved as a fixed point instead of an in-place raster sweep.
//
// The raster sweep computes new[r][c] = F(old[C,R,DR,D,DL], new[L,UL,U,UR]) -- block (r,c)
// sees the already-updated values of its left / upper neighbours and the old values of itself
// and its right / lower neighbours.  The dependency graph of `new` is acyclic, so the field
// is the unique fixed point of that system.  We reach it by:
//   pass 1   every block evaluated with new := old                          (k_reg_pass1)
//   pass 2   every block whose L/UL/U/UR changed in pass 1 is re-evaluated  (k_reg_pass2)
//   pass k   blocks pushed onto a work list by a changed neighbour          (k_reg_fix, k_reg_tail)
// until a pass changes nothing.  A block that changes pushes R, DR, D, DL (its dependants).
// k_reg_tail is a single workgroup that loops to convergence, so the launch sequence is
// fixed and needs no host synchronisation.  Energies are float32 exactly as the reference's
// (SAD + lambda * mult * Smoothness, FLT_MAX for out-of-image candidates, first strict min).
//
// BS x BS blocks; LPB lanes cooperate on one block, one image row per lane.
// =======================================================================================
struct RegArgs {
    const uint8_t *image1, *image2;
    int width, height;
    int rows, cols;             // grid at this block size
    const mv_t *old_grid;       // values before the sweep
    int old_shift;              // 1 when old_grid is the parent grid (divide_blocks fused), else 0
    int old_cols;
    mv_t *est;                  // rows x cols, the field being solved
    float lambda_mult;          // lambda * (float)lambda_multiplier, computed as the reference does
    // work lists
    uint32_t *list0, *list1;    // block indices
    uint32_t *bits0, *bits1;    // "already queued" bitmaps, one per list
    uint32_t *counters;         // [0..2] list lengths (rotating), [3] passes run, [4] blocks re-evaluated,
                                // [5] sticky: a sweep hit the pass cap without converging
    int pass;                   // pass number of this launch (k_reg_fix) / first pass (k_reg_tail)
};

template <int BS> struct RegCfg {
    static constexpr int LPB = BS >= 4 ? (BS > 64 ? 64 : BS) : 1;   // lanes per block
    static constexpr int ROWS_PER_LANE = BS / LPB;                  // 2 for BS == 2, else 1
};

// candidate order of motion_framework.cpp:441-449: C, L, R, DR, UL, UR, U, D, DL as (drow, dcol)
static constexpr int kNbRow[9] = {0, 0, 0, 1, -1, -1, -1, 1, 1};
static constexpr int kNbCol[9] = {0, -1, 1, 1, -1, 1, 0, 0, -1};
// which of them a raster sweep has already updated when it reaches the block
#define BBME_NEW_MASK ((1u << 1) | (1u << 4) | (1u << 5) | (1u << 6))

template <bool COHERENT>
__device__ __forceinline__ mv_t load_est(const mv_t *p)
{
    if constexpr (COHERENT)   // tail kernel: written by other waves of this workgroup a pass ago
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        return *p;
}

// SAD of one BS-pixel row: image1 at (bx, y1) [aligned], image2 at (x2, y2) [any alignment].
template <int BS>
__device__ __forceinline__ uint32_t row_sad(const RegArgs &a, int bx, int y1, int x2, int y2)
{
    if constexpr (BS >= 4) {
        constexpr int NW = BS / 4;
        const uint32_t *p1 = reinterpret_cast<const uint32_t *>(a.image1 + (size_t)y1 * a.width + bx);
        const int ax = x2 & ~3, sh = x2 & 3;
        const uint32_t *p2 = reinterpret_cast<const uint32_t *>(a.image2 + (size_t)y2 * a.width + ax);
        // the dword after the last one is only needed when sh != 0, and then it lies inside the row
        uint32_t w[NW + 1];
#pragma unroll
        for (int q = 0; q < NW; ++q) w[q] = p2[q];
        w[NW] = sh ? p2[NW] : 0u;
        uint32_t sad = 0;
#pragma unroll
        for (int q = 0; q < NW; ++q)
            sad = __builtin_amdgcn_sad_u8(p1[q], __builtin_amdgcn_alignbyte(w[q + 1], w[q], sh), sad);
        return sad;
    } else {   // BS == 2: two pixels
        const uint8_t *p1 = a.image1 + (size_t)y1 * a.width + bx;
        const uint8_t *p2 = a.image2 + (size_t)y2 * a.width + x2;
        const uint32_t v1 = (uint32_t)p1[0] | ((uint32_t)p1[1] << 8);
        const uint32_t v2 = (uint32_t)p2[0] | ((uint32_t)p2[1] << 8);
        return __builtin_amdgcn_sad_u8(v1, v2, 0u);
    }
}

// Evaluate block (r, c): returns the winning candidate MV.  All LPB lanes of the group call it
// with the same (r, c); `sub` is the lane's row inside the block.  use_new = bit mask of the
// candidates read from `est` instead of `old_grid`.
template <int BS, bool COHERENT>
__device__ __forceinline__ mv_t eval_block(const RegArgs &a, int r, int c, int sub, uint32_t use_new)
{
#pragma clang fp contract(off)
    constexpr int LPB = RegCfg<BS>::LPB;
    mv_t cand[9];
    uint32_t present = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int rr = r + kNbRow[k], cc = c + kNbCol[k];
        cand[k] = 0;
        if (rr >= 0 && rr < a.rows && cc >= 0 && cc < a.cols) {
            present |= 1u << k;
            if ((use_new >> k) & 1u)
                cand[k] = load_est<COHERENT>(a.est + (size_t)rr * a.cols + cc);
            else
                cand[k] = a.old_grid[(size_t)(rr >> a.old_shift) * a.old_cols + (cc >> a.old_shift)];
        }
    }
    const int bx = c * BS, by = r * BS;
    float energy[9];
    uint32_t inside = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        uint32_t sad = 0;
        if ((present >> k) & 1u) {
            const int x2 = bx + mv_x(cand[k]), y2 = by + mv_y(cand[k]);
            if (!(x2 < 0 || x2 > a.width - BS || y2 < 0 || y2 > a.height - BS)) {   // :578
                inside |= 1u << k;
                if constexpr (BS >= 4) {
#pragma unroll
                    for (int rw = sub; rw < BS; rw += LPB)
                        sad += row_sad<BS>(a, bx, by + rw, x2, y2 + rw);
                } else {
                    sad = row_sad<BS>(a, bx, by, x2, y2) + row_sad<BS>(a, bx, by + 1, x2, y2 + 1);
                }
            }
        }
        if constexpr (LPB > 1) {
#pragma unroll
            for (int o = LPB / 2; o > 0; o >>= 1) sad += __shfl_xor(sad, o);
        }
        energy[k] = (float)sad;
    }
    // calculate_smoothness (:623-644) with v_sad_u16 on bias-shifted halves: |u_m-u_k| + |v_m-v_k|
    int best = -1;
    float best_e = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if (!((present >> k) & 1u)) continue;
        float e = 3.402823466e+38f;                                              // FLT_MAX :580
        if ((inside >> k) & 1u) {
            const uint32_t ck = cand[k] ^ 0x80008000u;
            uint32_t smooth = 0;
#pragma unroll
            for (int m = 0; m < 9; ++m)
                if ((present >> m) & 1u)
                    smooth = __builtin_amdgcn_sad_u16(cand[m] ^ 0x80008000u, ck, smooth);
            const float t = a.lambda_mult * (float)smooth;
            e = energy[k] + t;                                                   // :607
        }
        if (best < 0 || e < best_e) { best = k; best_e = e; }                   // first strict min :648-660
    }
    mv_t res = cand[0];
#pragma unroll
    for (int k = 1; k < 9; ++k) if (best == k) res = cand[k];
    return res;
}

// dependants of block x = (r, c): R, DR, D, DL.  Queue them (once) on the next list.
__device__ __forceinline__ void push_dependants(const RegArgs &a, int r, int c,
                                                uint32_t *list, uint32_t *bits, uint32_t *count)
{
    const int dr[4] = {0, 1, 1, 1}, dc[4] = {1, 1, 0, -1};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int rr = r + dr[d], cc = c + dc[d];
        if (rr < 0 || rr >= a.rows || cc < 0 || cc >= a.cols) continue;
        const uint32_t x = (uint32_t)rr * a.cols + cc;
        const uint32_t bit = 1u << (x & 31);
        if (!(atomicOr(&bits[x >> 5], bit) & bit))
            list[atomicAdd(count, 1u)] = x;
    }
}

template <int BS>
__global__ __launch_bounds__(256) void k_reg_pass1(RegArgs a)
{
    constexpr int LPB = RegCfg<BS>::LPB;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t == 0) { a.counters[0] = 0; a.counters[1] = 0; a.counters[2] = 0; a.counters[3] = 1; a.counters[4] = 0; }
    const long long g = t / LPB;
    const int sub = (int)(t % LPB);
    if (g >= (long long)a.rows * a.cols) return;   // whole groups drop out together (LPB | 256)
    const int r = (int)(g / a.cols), c = (int)(g % a.cols);
    const mv_t res = eval_block<BS, false>(a, r, c, sub, 0u);
    if (sub == 0) a.est[g] = res;
}

// pass 2: pull form.  A block is re-evaluated iff one of its already-updated inputs differs
// from the old value pass 1 assumed.  Changes are written in place and push dependants for
// pass 3 (list1 / bits1 / counters[0]).
template <int BS>
__global__ __launch_bounds__(256) void k_reg_pass2(RegArgs a)
{
    constexpr int LPB = RegCfg<BS>::LPB;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long g = t / LPB;
    const int sub = (int)(t % LPB);
    if (g >= (long long)a.rows * a.cols) return;
    const int r = (int)(g / a.cols), c = (int)(g % a.cols);
    bool stale = false;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if (!((BBME_NEW_MASK >> k) & 1u)) continue;
        const int rr = r + kNbRow[k], cc = c + kNbCol[k];
        if (rr < 0 || rr >= a.rows || cc < 0 || cc >= a.cols) continue;
        const mv_t e = a.est[(size_t)rr * a.cols + cc];
        const mv_t o = a.old_grid[(size_t)(rr >> a.old_shift) * a.old_cols + (cc >> a.old_shift)];
        stale |= (e != o);
    }
    if (!stale) return;                              // uniform within the group
    const mv_t res = eval_block<BS, false>(a, r, c, sub, BBME_NEW_MASK);
    if (sub == 0) {
        atomicAdd(&a.counters[4], 1u);
        if (res != a.est[g]) {
            a.est[g] = res;
            push_dependants(a, r, c, a.list1, a.bits1, &a.counters[0]);
        }
    }
}

// one work-list pass; pass number p reads list[p&1] (length counters[p%3]), appends to
// list[(p+1)&1] (counters[(p+1)%3]) and zeroes counters[(p+2)%3] for the pass after.
template <int BS, bool COHERENT>
__device__ __forceinline__ void worklist_pass(const RegArgs &a, int p, uint32_t n,
                                              int group, int ngroups, int sub)
{
    const uint32_t *lcur = (p & 1) ? a.list1 : a.list0;
    uint32_t *bcur = (p & 1) ? a.bits1 : a.bits0;
    uint32_t *lnext = (p & 1) ? a.list0 : a.list1;
    uint32_t *bnext = (p & 1) ? a.bits0 : a.bits1;
    uint32_t *cnext = &a.counters[(p + 1) % 3];
    for (uint32_t idx = group; idx < n; idx += ngroups) {
        const uint32_t x = COHERENT ? __hip_atomic_load(&lcur[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                    : lcur[idx];
        const int r = (int)(x / a.cols), c = (int)(x % a.cols);
        if (sub == 0) atomicAnd(&bcur[x >> 5], ~(1u << (x & 31)));
        const mv_t res = eval_block<BS, COHERENT>(a, r, c, sub, BBME_NEW_MASK);
        if (sub == 0) {
            const mv_t prev = load_est<COHERENT>(a.est + x);
            if (res != prev) {
                if constexpr (COHERENT)
                    __hip_atomic_store(a.est + x, res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else
                    a.est[x] = res;
                push_dependants(a, r, c, lnext, bnext, cnext);
            }
        }
    }
}

template <int BS>
__global__ __launch_bounds__(256) void k_reg_fix(RegArgs a)
{
    constexpr int LPB = RegCfg<BS>::LPB;
    const int p = a.pass;
    const uint32_t n = a.counters[p % 3];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.counters[(p + 2) % 3] = 0;
        if (n) { a.counters[3] = p; atomicAdd(&a.counters[4], n); }
    }
    if (n == 0) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    worklist_pass<BS, false>(a, p, n, t / LPB, (int)(gridDim.x * 256) / LPB, t % LPB);
}

// Single workgroup, loops over work-list passes until one leaves the next list empty.
// Visibility between passes: every wave drains its stores (s_waitcnt vmcnt(0)) before the
// barrier; estimate / list / counter reads in the next pass bypass the L1 (agent-scope loads).
template <int BS>
__global__ __launch_bounds__(1024) void k_reg_tail(RegArgs a)
{
    constexpr int LPB = RegCfg<BS>::LPB;
    const int t = threadIdx.x;
    int p = a.pass;
    // a change can only travel along the raster dependency chain, whose length is below
    // 2*rows + cols; the cap is an exit every wave reaches even if that reasoning were wrong
    const int p_max = a.pass + 2 * a.rows + a.cols + 16;
    for (;;) {
        const uint32_t n = __hip_atomic_load(&a.counters[p % 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n == 0) break;
        if (p > p_max) { if (t == 0) a.counters[5] = 1; break; }      // reported by bbme_synchronize
        __syncthreads();                              // everyone has read n before it can be reused
        if (t == 0) {
            __hip_atomic_store(&a.counters[(p + 2) % 3], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.counters[3] = p;
            a.counters[4] += n;
        }
        worklist_pass<BS, true>(a, p, n, t / LPB, 1024 / LPB, t % LPB);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ++p;
    }
}

// =======================================================================================
// K5: MF::copy_to_all_pixels after the last divide (motion_framework.cpp:205-206, 815-826):
// every pixel of level 0 takes the MV of its 2x2 cell, as float2.  One thread per cell row pair.
// =======================================================================================
__global__ __launch_bounds__(256) void k_expand(const mv_t *cells, int cell_cols, int cell_rows,
                                                float *flow, int width)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)cell_cols * cell_rows * 2;     // two pixel rows per cell row
    if (t >= total) return;
    const int cx = (int)(t % cell_cols);
    const int y = (int)(t / cell_cols);                               // pixel row
    const mv_t m = cells[(size_t)(y >> 1) * cell_cols + cx];
    const float u = (float)mv_x(m), v = (float)mv_y(m);
    float4 o = make_float4(u, v, u, v);
    *reinterpret_cast<float4 *>(flow + 2 * ((size_t)y * width + 2 * cx)) = o;
}

// =======================================================================================
// MF::MF on the GPU (motion_framework.cpp:57-61, 86-106): zero border and pyrDown cascade.
// =======================================================================================
__global__ __launch_bounds__(256) void k_pad_zero(const uint8_t *src, int width, int height, int pitch,
                                                  int pad_x, int pad_y, uint8_t *dst, int pw, int ph)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;    // one output dword
    const int dwpr = pw / 4;
    if (t >= (long long)dwpr * ph) return;
    const int y = (int)(t / dwpr), x0 = (int)(t % dwpr) * 4;
    uint32_t w = 0;
    const int sy = y - pad_y;
    if (sy >= 0 && sy < height) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int sx = x0 + b - pad_x;
            if (sx >= 0 && sx < width) w |= (uint32_t)src[(size_t)sy * pitch + sx] << (8 * b);
        }
    }
    *reinterpret_cast<uint32_t *>(dst + (size_t)y * pw + x0) = w;
}

__device__ __forceinline__ int mirror101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

__global__ __launch_bounds__(256) void k_pyr_down(const uint8_t *src, int sw, int sh, uint8_t *dst)
{
    const int dw = sw / 2, dh = sh / 2;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)dw * dh) return;
    const int x = (int)(t % dw), y = (int)(t / dw);
    const int wgt[5] = {1, 4, 6, 4, 1};
    int xs[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) xs[k] = mirror101(2 * x + k - 2, sw);
    int acc = 0;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
        const uint8_t *row = src + (size_t)mirror101(2 * y + ky - 2, sh) * sw;
        int h = 0;
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) h += wgt[kx] * row[xs[kx]];
        acc += wgt[ky] * h;
    }
    dst[(size_t)y * dw + x] = (uint8_t)((acc + 128) >> 8);
}

// =======================================================================================
// instruction probes (bbme_selftest_isa)
// =======================================================================================
__global__ void k_probe_sad(const uint32_t *a, const uint32_t *b, const uint32_t *c,
                            uint32_t *sad_out, unsigned long long *qsad_out, uint32_t *align_out,
                            uint32_t *sad16_out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    sad_out[i] = __builtin_amdgcn_sad_u8(a[i], b[i], c[i]);
    const unsigned long long src0 = ((unsigned long long)b[i] << 32) | a[i];
    const unsigned long long acc = ((unsigned long long)(c[i] & 0x00ff00ffu) << 32) | (c[i] & 0x0f0f0f0fu);
    qsad_out[i] = __builtin_amdgcn_qsad_pk_u16_u8(src0, c[i] ^ a[i], acc);
    align_out[i] = __builtin_amdgcn_alignbyte(b[i], a[i], c[i] & 3u);
    sad16_out[i] = __builtin_amdgcn_sad_u16(a[i], b[i], c[i] & 0xffffu);
}

}  // namespace bbme
