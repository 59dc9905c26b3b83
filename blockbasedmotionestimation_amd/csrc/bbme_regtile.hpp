// bbme_regtile.hpp -- tile-resident regulariser kernels for gfx950.  Included by bbme_device.hip after
// bbme_kernels.hpp (uses RegArgs, score_block, dpp_row, mv_t).
//
// MF::regularize_MVs / find_min_candidate / calculate_smoothness / min_energy_candidate
// (motion_framework.cpp:424-662).  The raster sweep's result is the unique fixed point of
//     new[r][c] = F(old[C, R, DR, D, DL], new[L, UL, U, UR])
// (bbme_kernels.hpp, "K2").  A chain of dependent re-evaluations is the critical path of a sweep, and a
// chain step that goes through global memory costs four memory trips (candidates, image rows, store +
// drain, ownership atomics: ~2.3 us).  Here a workgroup owns a TILE of T x T blocks and keeps the tile's
// estimates and old values (+ a ring of neighbours) in LDS; a chain step inside a tile is one image-row
// trip plus ~200 instructions.
//
//   k_reg_tile   one workgroup per tile: pass 1 (every block with new := old), then the tile's local fixed
//                point -- four waves, each owning a band of rows, asynchronous, queues and dedup bitmap in
//                LDS -- under the assumption that the estimates of the blocks OUTSIDE the tile equal their
//                old values.  At the end the tile's estimates go to HBM, and every block outside the tile
//                that reads a changed border block as an already-updated input is appended to a compact
//                work list.
//   k_reg_solve  (bbme_kernels.hpp) takes that list instead of scanning a flag map and finishes the sweep
//                exactly: any block whose assumed inputs were wrong is on the list, and the solver re-
//                evaluates until nothing changes.
#pragma once

#include "bbme_kernels.hpp"

namespace bbme {

template <int BS> struct RegTile {
    static constexpr int T = BS <= 4 ? 32 : (BS == 8 ? 16 : 8);        // tile edge in blocks
    static constexpr int TP = T + 2;                                    // LDS pitch: one halo column each side
    static constexpr int NWAVES = 4;
    static constexpr int RPW = T / NWAVES;                              // rows of the tile a wave owns
    static constexpr int QCAP = T * T / NWAVES;                         // a wave's queue: its blocks, each at most once
    static constexpr int TSHIFT = T == 32 ? 5 : (T == 16 ? 4 : 3);
    static constexpr int MARKS = 8 * T;                                 // most border marks of one tile (6T + corners)
};

// Lane k16 of a 16-lane DPP row holds candidate k16's MV (order C, L, R, DR, UL, UR, U, D, DL, :441-449; lanes
// 9..15 absent).  Returns true in exactly one lane of the row: the candidate find_min_candidate picks -- lowest
// energy SAD + lambda * mult * Smooth (float32, :607; FLT_MAX outside the image, :578-582), first one among
// equals (:648-660).  All 16 lanes of the row must be active.
template <int BS>
__device__ __forceinline__ bool lanes_pick(const RegArgs &a, int bx, int by, int k16, bool present, mv_t mv)
{
#pragma clang fp contract(off)
    constexpr int NW = BS >= 4 ? BS / 4 : 1;
    struct __attribute__((packed, aligned(1))) row_t { uint32_t v[NW]; };
    constexpr uint32_t kMask = BS >= 4 ? 0xffffffffu : 0x0000ffffu;
    int x2 = bx + mv_x(mv), y2 = by + mv_y(mv);
    const bool inside = present && !(x2 < 0 || x2 > a.width - BS || y2 < 0 || y2 > a.height - BS);   // :578
    if (!inside) { x2 = bx; y2 = by; }
    uint32_t sad0 = 0, sad1 = 0;
    const uint8_t *p1 = a.image1 + (size_t)by * a.width + bx;
    const uint8_t *p2 = a.image2 + (size_t)y2 * a.width + x2;
    row_t u[BS], w[BS];
#pragma unroll
    for (int row = 0; row < BS; ++row) {
        u[row] = *reinterpret_cast<const row_t *>(p1 + (size_t)row * a.width);
        w[row] = *reinterpret_cast<const row_t *>(p2 + (size_t)row * a.width);
    }
    // smoothness while the rows are in flight: sum over the present candidates of |u_m - u_k| + |v_m - v_k|
    // (:637-641); candidate m reaches every lane of the row by row_newbcast
    const int lane = (int)(threadIdx.x & 63u);
    const int base = lane & ~15;
    const uint32_t pmask = (uint32_t)(__ballot(present) >> base) & 0x1ffu;
    const uint32_t mine = mv ^ 0x80008000u;
    uint32_t sm0 = 0, sm1 = 0;
#define BBME_SMOOTH_TERM(m, acc) \
    { const uint32_t other = dpp_row<0x150 + (m)>(mine); if ((pmask >> (m)) & 1u) acc = __builtin_amdgcn_sad_u16(other, mine, acc); }
    BBME_SMOOTH_TERM(0, sm0) BBME_SMOOTH_TERM(1, sm1) BBME_SMOOTH_TERM(2, sm0) BBME_SMOOTH_TERM(3, sm1) BBME_SMOOTH_TERM(4, sm0)
    BBME_SMOOTH_TERM(5, sm1) BBME_SMOOTH_TERM(6, sm0) BBME_SMOOTH_TERM(7, sm1) BBME_SMOOTH_TERM(8, sm0)
#undef BBME_SMOOTH_TERM
    const float t = a.lambda_mult * (float)(sm0 + sm1);
#pragma unroll
    for (int row = 0; row < BS; ++row)
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            if ((row + q) & 1) sad1 = __builtin_amdgcn_sad_u8(u[row].v[q] & kMask, w[row].v[q] & kMask, sad1);
            else sad0 = __builtin_amdgcn_sad_u8(u[row].v[q] & kMask, w[row].v[q] & kMask, sad0);
        }
    float e = 3.402823466e+38f;                                                 // FLT_MAX :580
    if (inside) e = (float)(sad0 + sad1) + t;                                   // :607
    // energies are >= 0, so their bit patterns order like the floats; absent lanes sort last
    const uint32_t ebits = present ? __float_as_uint(e) : 0xffffffffu;
    uint32_t m = ebits;
    m = min(m, dpp_row<0xB1>(m));        // quad_perm [1,0,3,2]
    m = min(m, dpp_row<0x4E>(m));        // quad_perm [2,3,0,1]
    m = min(m, dpp_row<0x141>(m));       // row_half_mirror
    m = min(m, dpp_row<0x140>(m));       // row_mirror: every lane of the row holds the row minimum
    const uint32_t eq = (uint32_t)(__ballot(ebits == m) >> base) & 0xffffu;    // never empty: candidate 0 is present
    return k16 == __builtin_ctz(eq);
}

// (drow + 1) and (dcol + 1) of candidate k, two bits each, order C,L,R,DR,UL,UR,U,D,DL (:441-449)
static constexpr uint32_t kCandRowCode = 1u | 1u << 2 | 1u << 4 | 2u << 6 | 0u << 8 | 0u << 10 | 0u << 12 | 2u << 14 | 2u << 16;
static constexpr uint32_t kCandColCode = 1u | 0u << 2 | 2u << 4 | 2u << 6 | 0u << 8 | 2u << 10 | 1u << 12 | 1u << 14 | 0u << 16;

// LDS state of one tile (k_reg_tile)
template <int BS> struct TileState {
    using C = RegTile<BS>;
    mv_t old_t[C::TP * C::TP];            // values before the sweep, tile + ring
    mv_t est_t[C::TP * C::TP];            // the field being solved; ring = the assumption about the neighbours
    uint32_t q[C::NWAVES][C::QCAP];       // per-wave queue of (local block index + 1); 0 = slot not written yet
    uint32_t queued[C::T * C::T / 32];    // a block is on a queue
    uint32_t tail[C::NWAVES];
    uint32_t pending;                     // blocks queued or being evaluated, tile-wide
    uint32_t n_marks, mark_base;
};

#define BBME_LDS_ORDER() asm volatile("" ::: "memory")      // DS instructions of a wave execute in issue order

// Block `bit` (local index) of the tile has a stale input: put it on its owner's queue unless it is there already.
template <int BS>
__device__ __forceinline__ void tile_push(TileState<BS> &s, uint32_t bit)
{
    using C = RegTile<BS>;
    const uint32_t mbit = 1u << (bit & 31u);
    if (atomicOr(&s.queued[bit >> 5], mbit) & mbit) return;            // its owner has not looked at its inputs yet
    const uint32_t w2 = (bit >> C::TSHIFT) / C::RPW;
    atomicAdd(&s.pending, 1u);                                          // before the entry becomes visible
    const uint32_t pos = atomicAdd(&s.tail[w2], 1u);
    s.q[w2][pos & (C::QCAP - 1)] = bit + 1u;
}

// The local fixed point: every wave evaluates the queued blocks of its band of rows, four per step (16 lanes per
// block, lane k = candidate k), until no block of the tile is queued or being evaluated.
template <int BS>
__device__ __forceinline__ void tile_solve(const RegArgs &a, TileState<BS> &s, int r0, int c0, uint32_t round_cap,
                                           uint32_t &evaluated)
{
    using C = RegTile<BS>;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = lane >> 4, k16 = lane & 15;
    const int k = k16 < 9 ? k16 : 0;
    const int dr = (int)((kCandRowCode >> (2 * k)) & 3u) - 1, dc = (int)((kCandColCode >> (2 * k)) & 3u) - 1;
    const bool from_est = (BBME_NEW_MASK >> k) & 1u;
    uint32_t head = 0, spins = 0, steps = 0;
    for (;;) {
        BBME_LDS_ORDER();
        const uint32_t tl = __builtin_amdgcn_readfirstlane(*(volatile uint32_t *)&s.tail[w]);
        const uint32_t pend = __builtin_amdgcn_readfirstlane(*(volatile uint32_t *)&s.pending);
        uint32_t cnt = min(tl - head, 4u);
        if (cnt == 0) {
            if (pend == 0) break;
            // every spin either ends with work or with pending == 0; the cap is an exit should that ever be wrong
            if (++spins > round_cap) { if (lane == 0) a.counters[5] = 1; break; }
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        const uint32_t slot = (head + (uint32_t)g) & (C::QCAP - 1);
        uint32_t v = (uint32_t)g < cnt ? *(volatile uint32_t *)&s.q[w][slot] : 0u;
        // an entry whose tail increment is visible may not have been written yet: take the leading written ones
        const unsigned long long rdy = __ballot(v != 0);
        const uint32_t nr = (rdy & 1ull) ? ((rdy >> 16 & 1ull) ? ((rdy >> 32 & 1ull) ? ((rdy >> 48 & 1ull) ? 4u : 3u) : 2u) : 1u) : 0u;
        cnt = min(cnt, nr);
        if (cnt == 0 || ++steps > round_cap) { if (++spins > round_cap || steps > round_cap) { if (lane == 0) a.counters[5] = 1; break; } continue; }
        const bool active = (uint32_t)g < cnt;
        if (!active) v = 1;                                            // idle rows shadow block 0 (never stored)
        const uint32_t bit = v - 1u;
        const int lr = (int)(bit >> C::TSHIFT), lc = (int)(bit & (C::T - 1));
        if (active && k16 == 0) {                                       // off the queue BEFORE its inputs are read
            s.q[w][slot] = 0;
            atomicAnd(&s.queued[bit >> 5], ~(1u << (bit & 31u)));
        }
        head += cnt;
        BBME_LDS_ORDER();
        const int r = r0 + lr, c = c0 + lc;
        const int rr = r + dr, cc = c + dc;
        const bool present = active && k16 < 9 && rr >= 0 && rr < a.rows && cc >= 0 && cc < a.cols;
        const int li = (lr + dr + 1) * C::TP + lc + dc + 1;
        const mv_t mv = from_est ? *(volatile mv_t *)&s.est_t[li] : s.old_t[li];
        const mv_t prev = *(volatile mv_t *)&s.est_t[(lr + 1) * C::TP + lc + 1];
        const bool win = lanes_pick<BS>(a, c * BS, r * BS, k16, present, mv);
        if (active && win && mv != prev) {
            *(volatile mv_t *)&s.est_t[(lr + 1) * C::TP + lc + 1] = mv;
            BBME_LDS_ORDER();
            // dependants R, DR, D, DL inside the tile (those outside are told at the end of the kernel): the four
            // dedup atomics first, then the queue operations of the ones that were not queued yet, so that the LDS
            // round trips overlap
            const bool right = lc + 1 < C::T && c + 1 < a.cols, down = lr + 1 < C::T && r + 1 < a.rows;
            const bool want[4] = {right, down && right, down, down && lc > 0};
            const uint32_t dbit[4] = {bit + 1u, bit + C::T + 1u, bit + C::T, bit + C::T - 1u};
            uint32_t was[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                was[d] = ~0u;
                if (want[d]) was[d] = atomicOr(&s.queued[dbit[d] >> 5], 1u << (dbit[d] & 31u));
            }
            bool fresh[4];
            uint32_t nfresh = 0;
#pragma unroll
            for (int d = 0; d < 4; ++d) { fresh[d] = !((was[d] >> (dbit[d] & 31u)) & 1u); nfresh += fresh[d]; }
            if (nfresh) atomicAdd(&s.pending, nfresh);                  // before the entries become visible
            BBME_LDS_ORDER();
            uint32_t pos[4];
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (fresh[d]) pos[d] = atomicAdd(&s.tail[(dbit[d] >> C::TSHIFT) / C::RPW], 1u);
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (fresh[d]) s.q[(dbit[d] >> C::TSHIFT) / C::RPW][pos[d] & (C::QCAP - 1)] = dbit[d] + 1u;
        }
        BBME_LDS_ORDER();
        if (lane == 0) atomicSub(&s.pending, cnt);                      // after this step's own pushes
        evaluated += cnt;
        spins = 0;
    }
}

template <int BS>
__global__ __launch_bounds__(256) void k_reg_tile(RegArgs a)
{
    using C = RegTile<BS>;
    constexpr int T = C::T, TP = C::TP;
    constexpr int LPB = RegCfg<BS>::LPB;
    __shared__ TileState<BS> s;
    const int t = threadIdx.x;
    const int tiles_x = (a.cols + T - 1) / T;
    const int r0 = ((int)blockIdx.x / tiles_x) * T, c0 = ((int)blockIdx.x % tiles_x) * T;
    if (blockIdx.x == 0 && t == 0) {
        a.counters[3] = 0; a.counters[4] = 0; a.counters[6] = 0; a.counters[7] = 0; a.counters[8] = 0;
    }
    // ---- the tile and its ring: old values; the estimates start as the old values -------------------
    for (int i = t; i < TP * TP; i += 256) {
        const int rr = r0 - 1 + i / TP, cc = c0 - 1 + i % TP;
        mv_t o = 0;
        if (rr >= 0 && rr < a.rows && cc >= 0 && cc < a.cols)
            o = a.old_grid[(size_t)(rr >> a.old_shift) * a.old_cols + (cc >> a.old_shift)];
        s.old_t[i] = o;
        s.est_t[i] = o;
    }
    for (int i = t; i < C::NWAVES * C::QCAP; i += 256) (&s.q[0][0])[i] = 0;
    if (t < T * T / 32) s.queued[t] = 0;
    if (t < C::NWAVES) s.tail[t] = 0;
    if (t == 0) { s.pending = 0; s.n_marks = 0; }
    __syncthreads();
    // ---- pass 1: every block with new := old ----------------------------------------------------------
    {
        const int sub = t % LPB;
        for (int idx = t / LPB; idx < T * T; idx += 256 / LPB) {
            const int lr = idx >> C::TSHIFT, lc = idx & (T - 1);
            const int r = r0 + lr, c = c0 + lc;
            if (r >= a.rows || c >= a.cols) continue;                   // whole groups skip together
            mv_t cand[9];
            uint32_t present = 0;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int rr = r + kNbRow[k], cc = c + kNbCol[k];
                if (rr >= 0 && rr < a.rows && cc >= 0 && cc < a.cols) present |= 1u << k;
                cand[k] = s.old_t[(lr + kNbRow[k] + 1) * TP + lc + kNbCol[k] + 1];
            }
            bool uniform = true;
#pragma unroll
            for (int k = 1; k < 9; ++k) uniform &= !((present >> k) & 1u) || cand[k] == cand[0];
            mv_t res = cand[0];
            if (!uniform) res = score_block<BS, true>(a, cand, present, c * BS, r * BS, sub);
            if (sub == 0) s.est_t[(lr + 1) * TP + lc + 1] = res;
        }
    }
    __syncthreads();
    // ---- blocks whose already-updated inputs inside the tile came out different from their old values ---
    for (int idx = t; idx < T * T; idx += 256) {
        const int lr = idx >> C::TSHIFT, lc = idx & (T - 1);
        if (r0 + lr >= a.rows || c0 + lc >= a.cols) continue;
        bool stale = false;
        if (lc > 0) stale |= s.est_t[(lr + 1) * TP + lc] != s.old_t[(lr + 1) * TP + lc];                       // L
        if (lr > 0) {
            stale |= s.est_t[lr * TP + lc + 1] != s.old_t[lr * TP + lc + 1];                                   // U
            if (lc > 0) stale |= s.est_t[lr * TP + lc] != s.old_t[lr * TP + lc];                               // UL
            if (lc + 1 < T && c0 + lc + 1 < a.cols) stale |= s.est_t[lr * TP + lc + 2] != s.old_t[lr * TP + lc + 2];   // UR
        }
        if (stale) tile_push<BS>(s, (uint32_t)idx);
    }
    __syncthreads();
    uint32_t evaluated = 0;
    tile_solve<BS>(a, s, r0, c0, a.round_cap, evaluated);
    if ((t & 63) == 0 && evaluated) atomicAdd(&a.counters[4], evaluated);
    __syncthreads();
    // ---- the tile's estimates to HBM; tell the blocks outside that read a changed border block ----------
    for (int idx = t; idx < T * T; idx += 256) {
        const int lr = idx >> C::TSHIFT, lc = idx & (T - 1);
        const int r = r0 + lr, c = c0 + lc;
        if (r >= a.rows || c >= a.cols) continue;
        const mv_t e = s.est_t[(lr + 1) * TP + lc + 1];
        a.est[(size_t)r * a.cols + c] = e;
        if (e == s.old_t[(lr + 1) * TP + lc + 1]) continue;
        const int ddr[4] = {0, 1, 1, 1}, ddc[4] = {1, 1, 0, -1};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int lr2 = lr + ddr[d], lc2 = lc + ddc[d], rr = r + ddr[d], cc = c + ddc[d];
            if (rr >= a.rows || cc < 0 || cc >= a.cols) continue;
            if (lr2 < T && lc2 >= 0 && lc2 < T) continue;                // inside: the local fixed point saw it
            const uint32_t pos = atomicAdd(&s.n_marks, 1u);
            (&s.q[0][0])[pos] = (uint32_t)rr * a.cols + cc;              // the queues are idle now
        }
    }
    __syncthreads();
    const uint32_t n = s.n_marks;
    if (n == 0) return;
    if (t == 0) {
        s.mark_base = atomicAdd(a.mark_count, n);
        if (s.mark_base + n > a.mark_cap) a.counters[5] = 1;            // cannot happen (<= 6T marks per tile); refuse the result
    }
    __syncthreads();
    if (s.mark_base + n > a.mark_cap) return;
    for (uint32_t i = t; i < n; i += 256) a.mark_list[s.mark_base + i] = (&s.q[0][0])[i];
}

}  // namespace bbme
