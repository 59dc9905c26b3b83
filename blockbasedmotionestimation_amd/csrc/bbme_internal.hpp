// bbme_internal.hpp -- declarations shared by the host side and the HIP side of libbbme.so.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "bbme.h"

namespace bbme {

// ---- error channel -------------------------------------------------------------------
int fail(int status, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();

// ---- host prep (MF::MF, motion_framework.cpp:4-111) ----------------------------------
struct Geometry {
    int width = 0, height = 0;             // original frame
    int padded_width = 0, padded_height = 0, pad_x = 0, pad_y = 0;
};
int plan_padding(int width, int height, const bbme_params &p, Geometry &g);
int validate_params(const bbme_params &p);
void pad_zero(const uint8_t *src, int width, int height, int pitch, int pad_x, int pad_y, uint8_t *dst);
void pyr_down(const uint8_t *src, int sw, int sh, uint8_t *dst);
void resize_x4(const uint8_t *src, int sw, int sh, uint8_t *dst);

// ---- spiral order of find_min_block_spiral (motion_framework.cpp:326-411) ------------
// Built by walking the spiral exactly as the reference loop does.
struct SpiralTable {
    int range = 0;                         // R = max(0, (search_size - block_size) >> 1)
    int side = 0;                          // 2R+1
    std::vector<int16_t> dx, dy;           // visit order: rank -> (dx, dy)
    std::vector<uint16_t> rank_of;         // [(dy+R)*rank_pitch + (dx+R)] -> rank
    int rank_pitch = 0;                    // >= side, multiple of 4 (groups of 4 dx)
};
SpiralTable build_spiral(int search_size, int block_size);

// Work split of the fast search kernel: the (2R+1)^2 candidate square as column groups of 4 dx
// times vertical strips, packed into rounds of up to 64 strips of equal height.
struct SearchPlan {
    std::vector<uint32_t> rounds;          // strip height S of each round (16, 8, 4, 2 or 1)
    std::vector<uint32_t> tasks;           // rounds.size() * 64: g | dy0 << 8, 0xffffffff = idle lane
    int groups = 0;                        // column groups = ceil((2R+1) / 4)
    int pitch_dw = 0;                      // LDS window pitch in dwords (odd, >= groups + B/4)
};
// lanes = 64 x the waves that share one macroblock: a round is `lanes` tasks, wave w takes tasks [64 w, 64 w + 64)
SearchPlan plan_search(int range, int block_size, int max_strip, int lanes = 64);

// ---- Flow (rw_flow.cpp) ---------------------------------------------------------------
int flo_read(const char *filename, int *width, int *height, float **data);
int flo_write(const char *filename, int width, int height, const float *data);
double calculate_mse(const float *gtruth, const float *flow, int width, int height);
void motion_to_color(const float *flow, int width, int height, float maxmotion, uint8_t *bgr, float range[5]);
int ppm_write_bgr(const char *filename, int width, int height, const uint8_t *bgr);
void subsample_div4(const float *flow_padded, int padded_width, int padded_height,
                    int pad_x, int pad_y, float *out, int out_width);

}  // namespace bbme
