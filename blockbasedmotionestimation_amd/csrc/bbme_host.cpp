// bbme_host.cpp -- host side of libbbme.so: everything of the drop-in boundary that the
// reference also does on the CPU.  MF::MF's padding + pyramid (motion_framework.cpp:4-111),
// the spiral visiting order (:326-411) as a table, and the Flow class' .flo codec and EPE
// (rw_flow.cpp:39-200,309-332).  No GPU code here; no dependency on oracle/.
#include "bbme_internal.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <set>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <unistd.h>

namespace bbme {

// ---------------------------------------------------------------------------------------
// error channel
// ---------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

int fail(int status, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return status;
}

void clear_error() { g_last_error.clear(); }

static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

int validate_params(const bbme_params &p)
{
    if (p.num_levels <= 0 || p.num_levels > BBME_MAX_LEVELS)      // assert(num_levels > 0), :7
        return fail(BBME_ERR_INVALID, "num_levels must be in 1..%d (got %d)", BBME_MAX_LEVELS, p.num_levels);
    for (int i = 0; i < p.num_levels; ++i) {
        if (!is_pow2(p.block_size[i]) || p.block_size[i] < 2 || p.block_size[i] > 64)
            return fail(BBME_ERR_UNSUPPORTED,
                        "block_size[%d]=%d: the kernels need a power of two in 2..64 "
                        "(the regulariser halves blocks down to 2x2)", i, p.block_size[i]);
        if (p.search_size[i] <= 0)
            return fail(BBME_ERR_INVALID, "search_size[%d]=%d must be positive", i, p.search_size[i]);
        int range = (p.search_size[i] - p.block_size[i]) >> 1;
        // (ranks of the spiral walk are 16-bit: (2 * 127 + 1)^2 - 1 < 0xffff.  Ranges up to 63 take the strip kernel where the block
        // size allows, larger ones the generic kernel)
        if (range > 127)
            return fail(BBME_ERR_UNSUPPORTED, "search range %d at level %d exceeds 127", range, i);
    }
    return BBME_OK;
}

// ---------------------------------------------------------------------------------------
// Padding search (motion_framework.cpp:14-54).  The reference grows height and width one
// pixel at a time until both are divisible by 2^i * block_size[i] for every level, and
// gives up when either reaches twice the original.  Each dimension therefore ends at the
// next multiple of M = lcm_i(2^i * block_size[i]); reaching 2x the original is an error.
// ---------------------------------------------------------------------------------------
int plan_padding(int width, int height, const bbme_params &p, Geometry &g)
{
    if (width <= 0 || height <= 0)
        return fail(BBME_ERR_INVALID, "frame size %dx%d is not positive", width, height);
    long long m = 1;
    for (int i = 0; i < p.num_levels; ++i)
        m = std::lcm(m, (long long)p.block_size[i] << i);
    long long th = ((height + m - 1) / m) * m, tw = ((width + m - 1) / m) * m;
    if (th >= 2LL * height || tw >= 2LL * width)
        return fail(BBME_ERR_PADDING,
                    "Could not find any multiples of the block size that match padded image dimensions "
                    "(%dx%d, need multiples of %lld)", width, height, m);
    if ((th - height) % 2 != 0 || (tw - width) % 2 != 0)
        return fail(BBME_ERR_ODD_PADDING,
                    "padded size %lldx%lld differs from %dx%d by an odd amount: the reference would "
                    "allocate a mis-sized image (motion_framework.cpp:50-58)", tw, th, width, height);
    g.width = width; g.height = height;
    g.padded_width = (int)tw; g.padded_height = (int)th;
    g.pad_x = (int)(tw - width) / 2; g.pad_y = (int)(th - height) / 2;
    return BBME_OK;
}

void pad_zero(const uint8_t *src, int width, int height, int pitch, int pad_x, int pad_y, uint8_t *dst)
{
    const int pw = width + 2 * pad_x;
    uint8_t *out = dst;
    for (int y = 0; y < pad_y; ++y, out += pw) memset(out, 0, (size_t)pw);
    for (int y = 0; y < height; ++y, out += pw) {
        memset(out, 0, (size_t)pad_x);
        memcpy(out + pad_x, src + (size_t)y * pitch, (size_t)width);
        memset(out + pad_x + width, 0, (size_t)pad_x);
    }
    for (int y = 0; y < pad_y; ++y, out += pw) memset(out, 0, (size_t)pw);
}

// cv::pyrDown on 8-bit data (motion_framework.cpp:89-90): 5x5 binomial [1 4 6 4 1]^2 / 256,
// integer with one rounding (+128 >> 8), BORDER_REFLECT_101, destination pixel (x,y) centred on
// source (2x,2y).  Horizontal sums of the five source rows of a destination row are kept in a
// ring so every source row is filtered once.
static inline int mirror101(int p, int n)
{
    if (n == 1) return 0;
    for (;;) {
        if (p < 0) p = -p;
        else if (p >= n) p = 2 * n - 2 - p;
        else return p;
    }
}

void pyr_down(const uint8_t *src, int sw, int sh, uint8_t *dst)
{
    const int dw = sw / 2, dh = sh / 2;
    std::vector<int> hsum((size_t)sh * dw);
    std::vector<int> xm((size_t)dw * 5);
    for (int x = 0; x < dw; ++x)
        for (int t = 0; t < 5; ++t) xm[(size_t)x * 5 + t] = mirror101(2 * x + t - 2, sw);
    for (int y = 0; y < sh; ++y) {
        const uint8_t *s = src + (size_t)y * sw;
        int *h = &hsum[(size_t)y * dw];
        for (int x = 0; x < dw; ++x) {
            const int *m = &xm[(size_t)x * 5];
            h[x] = s[m[0]] + 4 * s[m[1]] + 6 * s[m[2]] + 4 * s[m[3]] + s[m[4]];
        }
    }
    for (int y = 0; y < dh; ++y) {
        const int *r0 = &hsum[(size_t)mirror101(2 * y - 2, sh) * dw];
        const int *r1 = &hsum[(size_t)mirror101(2 * y - 1, sh) * dw];
        const int *r2 = &hsum[(size_t)mirror101(2 * y, sh) * dw];
        const int *r3 = &hsum[(size_t)mirror101(2 * y + 1, sh) * dw];
        const int *r4 = &hsum[(size_t)mirror101(2 * y + 2, sh) * dw];
        uint8_t *d = dst + (size_t)y * dw;
        for (int x = 0; x < dw; ++x)
            d[x] = (uint8_t)((r0[x] + 4 * r1[x] + 6 * r2[x] + 4 * r3[x] + r4[x] + 128) >> 8);
    }
}

// cv::resize(img, img, Size(), 4, 4, INTER_LINEAR) on 8-bit data (main_class.cpp:32-33):
// OpenCV's fixed-point bilinear -- 11-bit coefficients, horizontal pass to int, vertical pass
// ((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2.
void resize_x4(const uint8_t *src, int sw, int sh, uint8_t *dst)
{
    const int dw = sw * 4, dh = sh * 4;
    auto coef = [](float f) {
        long r = lrintf(f * 2048.f);
        return (short)(r > 32767 ? 32767 : (r < -32768 ? -32768 : r));
    };
    std::vector<int> sx0(dw), sx1(dw);
    std::vector<short> ax0(dw), ax1(dw);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * 0.25 - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { sx = 0; fx = 0; }
        if (sx >= sw - 1) { sx = sw - 1; fx = 0; }
        sx0[dx] = sx;
        sx1[dx] = sx + 1 < sw ? sx + 1 : sx;       // weight is 0 there
        ax0[dx] = coef(1.f - fx);
        ax1[dx] = coef(fx);
        if (sx + 1 >= sw) { ax0[dx] = 2048; ax1[dx] = 0; }
    }
    std::vector<int> row0(dw), row1(dw);
    int have0 = -1, have1 = -1;
    auto hpass = [&](int sy, std::vector<int> &out) {
        const uint8_t *s = src + (size_t)sy * sw;
        for (int dx = 0; dx < dw; ++dx) out[dx] = s[sx0[dx]] * ax0[dx] + s[sx1[dx]] * ax1[dx];
    };
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * 0.25 - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        int b0 = coef(1.f - fy), b1 = coef(fy);
        int y0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
        int y1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
        if (have0 != y0) {
            if (have1 == y0) { row0.swap(row1); std::swap(have0, have1); }
            else { hpass(y0, row0); have0 = y0; }
        }
        if (have1 != y1) { hpass(y1, row1); have1 = y1; }
        uint8_t *d = dst + (size_t)dy * dw;
        for (int dx = 0; dx < dw; ++dx)
            d[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
}

// ---------------------------------------------------------------------------------------
// Spiral order (motion_framework.cpp:326-411): centre; then for m = 1,3,5,... < shift:
// right m, down m, left m+1, up m+1; finally right m-1.  Walked literally.
// ---------------------------------------------------------------------------------------
SpiralTable build_spiral(int search_size, int block_size)
{
    SpiralTable t;
    const int shift = search_size - block_size;
    int l = 0, k = 0, m, s;
    auto emit = [&]() { t.dx.push_back((int16_t)l); t.dy.push_back((int16_t)k); };
    emit();
    for (m = 1; m < shift; m += 2) {
        for (s = 0; s < m; ++s) { ++l; emit(); }
        for (s = 0; s < m; ++s) { ++k; emit(); }
        for (s = 0; s < m + 1; ++s) { --l; emit(); }
        for (s = 0; s < m + 1; ++s) { --k; emit(); }
    }
    for (s = 0; s < m - 1; ++s) { ++l; emit(); }
    int r = 0;
    for (size_t i = 0; i < t.dx.size(); ++i) {
        r = std::max(r, std::abs((int)t.dx[i]));
        r = std::max(r, std::abs((int)t.dy[i]));
    }
    t.range = r;
    t.side = 2 * r + 1;
    t.rank_pitch = (t.side + 3) & ~3;
    t.rank_of.assign((size_t)t.rank_pitch * t.side, 0xFFFF);
    for (size_t i = 0; i < t.dx.size(); ++i)
        t.rank_of[(size_t)(t.dy[i] + r) * t.rank_pitch + (t.dx[i] + r)] = (uint16_t)i;
    return t;
}

// ---------------------------------------------------------------------------------------
// Work split of k_search_fast.  Every (column group, candidate row) pair must be covered exactly
// once.  A round costs its strip height whatever the number of busy lanes, so rounds are kept
// full: take the tallest strip height that still has 64 strips available; what is left (< 64
// single rows) goes into one last round of height 1.
// ---------------------------------------------------------------------------------------
SearchPlan plan_search(int range, int block_size, int max_strip, int lanes)
{
    SearchPlan p;
    const int n = 2 * range + 1;
    p.groups = (n + 3) / 4;
    p.pitch_dw = (p.groups + block_size / 4) | 1;
    // n = 4 G' + 1 (every even range): the last column group would hold ONE candidate column and three of padding, and
    // the last candidate row a strip of height one in every group -- 9 % of the QSADs of a +-32 search.  The tight plan
    // packs the G' x (n - 1) rectangle into strips and gives the rim its own rounds (kernel: search_row_parts,
    // search_aligned_column): kind 1, candidate row n - 1 of the G' full groups, four lanes per (group, row), a quarter of
    // the block's rows each; kind 2, candidate column n - 1 (dx = +R, dword aligned in the staged window: R is even),
    // one candidate per lane, v_sad_u8.  rounds[] = S | kind << 8.
    static const bool loose = getenv("BBME_LOOSE_PLAN") != nullptr;   // read once: every context of a process plans alike
    const bool tight = n % 4 == 1 && n >= 9 && block_size <= 16 && !loose;
    const int groups_main = tight ? p.groups - 1 : p.groups, rows_main = tight ? n - 1 : n;
    std::vector<int> next(groups_main, 0);                   // first uncovered candidate row per column group
    auto emit_round = [&](int s, int want) {
        p.rounds.push_back((uint32_t)s);
        std::vector<uint32_t> t;
        // strips row by row of columns: lane = strip_row * groups + g keeps a round's LDS reads spread
        bool more = true;
        while (more && (int)t.size() < want) {
            more = false;
            for (int g = 0; g < groups_main && (int)t.size() < want; ++g)
                if (rows_main - next[g] >= s) {
                    t.push_back((uint32_t)g | ((uint32_t)next[g] << 8));
                    next[g] += s;
                    more = true;
                }
        }
        t.resize((size_t)((t.size() + lanes - 1) / lanes * lanes), 0xffffffffu);   // whole rounds of `lanes` tasks
        p.tasks.insert(p.tasks.end(), t.begin(), t.end());
    };
    for (int s = max_strip; s >= 1; s >>= 1) {
        for (;;) {
            int avail = 0;
            for (int g = 0; g < groups_main; ++g) avail += (rows_main - next[g]) / s;
            if (avail >= lanes) emit_round(s, lanes);
            else break;
        }
    }
    int left = 0;
    for (int g = 0; g < groups_main; ++g) left += rows_main - next[g];
    while (left > 0) { emit_round(1, std::min(left, lanes)); left -= std::min(left, lanes); }
    if (tight) {
        auto emit_tasks = [&](uint32_t kind, const std::vector<uint32_t> &all) {
            for (size_t i = 0; i < all.size(); i += (size_t)lanes) {
                p.rounds.push_back(1u | kind << 8);
                for (int k = 0; k < lanes; ++k)
                    p.tasks.push_back(i + k < all.size() ? all[i + k] : 0xffffffffu);
            }
        };
        std::vector<uint32_t> parts, column;
        for (int g = 0; g < groups_main; ++g)
            for (uint32_t part = 0; part < 4; ++part)       // the four lanes of a quad share one (group, row)
                parts.push_back((uint32_t)g | (uint32_t)(n - 1) << 8 | part << 16);
        for (int dy = 0; dy < n; ++dy) column.push_back((uint32_t)dy);
        emit_tasks(1u, parts);
        emit_tasks(2u, column);
    }
    return p;
}

// ---------------------------------------------------------------------------------------
// Flow::ReadFlowFile / WriteFlowFile / CalculateMSE (rw_flow.cpp)
// ---------------------------------------------------------------------------------------
static const float kTagFloat = 202021.25f;      // rw_flow.cpp:25
static const char kTagString[] = "PIEH";        // rw_flow.cpp:26

int flo_read(const char *filename, int *width, int *height, float **data)
{
    if (!filename) return fail(BBME_ERR_IO, "ReadFlowFile: empty filename");
    const char *dot = strrchr(filename, '.');
    if (!dot || strcmp(dot, ".flo") != 0)
        return fail(BBME_ERR_IO, "ReadFlowFile extension .flo expected");
    FILE *f = fopen(filename, "rb");
    if (!f) return fail(BBME_ERR_IO, "ReadFlowFile: could not open file %s", filename);
    float tag; int w, h;
    if (fread(&tag, 4, 1, f) != 1 || fread(&w, 4, 1, f) != 1 || fread(&h, 4, 1, f) != 1) {
        fclose(f); return fail(BBME_ERR_IO, "ReadFlowFile: problem reading file");
    }
    if (tag != kTagFloat) { fclose(f); return fail(BBME_ERR_IO, "ReadFlowFile: wrong tag (possibly due to big-endian machine?)"); }
    if (w < 1 || w > 99999) { fclose(f); return fail(BBME_ERR_IO, "ReadFlowFile: illegal width %d", w); }
    if (h < 1 || h > 99999) { fclose(f); return fail(BBME_ERR_IO, "ReadFlowFile: illegal height %d", h); }
    const size_t n = (size_t)w * h * 2;
    float *buf = (float *)malloc(n * sizeof(float));
    if (!buf) { fclose(f); return fail(BBME_ERR_IO, "ReadFlowFile: out of memory"); }
    if (fread(buf, sizeof(float), n, f) != n) { free(buf); fclose(f); return fail(BBME_ERR_IO, "ReadFlowFile: file is too short"); }
    if (fgetc(f) != EOF) { free(buf); fclose(f); return fail(BBME_ERR_IO, "ReadFlowFile: file is too long"); }
    fclose(f);
    *width = w; *height = h; *data = buf;
    return BBME_OK;
}

int flo_write(const char *filename, int width, int height, const float *data)
{
    if (!filename) return fail(BBME_ERR_IO, "WriteFlowFile: empty filename");
    const char *dot = strrchr(filename, '.');
    if (!dot) return fail(BBME_ERR_IO, "WriteFlowFile: extension required in filename");
    if (strcmp(dot, ".flo") != 0) return fail(BBME_ERR_IO, "WriteFlowFile: filename should have extension '.flo'");
    if (width < 1 || height < 1 || !data) return fail(BBME_ERR_INVALID, "WriteFlowFile: empty image");
    FILE *f = fopen(filename, "wb");
    if (!f) return fail(BBME_ERR_IO, "WriteFlowFile: could not open file %s", filename);
    const size_t n = (size_t)width * height * 2;
    bool ok = fwrite(kTagString, 1, 4, f) == 4 && fwrite(&width, 4, 1, f) == 1 && fwrite(&height, 4, 1, f) == 1;
    ok = ok && fwrite(data, sizeof(float), n, f) == n;
    ok = (fclose(f) == 0) && ok;
    return ok ? BBME_OK : fail(BBME_ERR_IO, "WriteFlowFile: problem writing data");
}

static inline bool unknown_flow(float u, float v)
{
    return std::fabs(u) > 1e9 || std::fabs(v) > 1e9 || std::isnan(u) || std::isnan(v);
}

double calculate_mse(const float *gtruth, const float *flow, int width, int height)
{
    long long count = 0;
    double error = 0;
    const size_t n = (size_t)width * height;
    for (size_t i = 0; i < n; ++i) {
        const float gu = gtruth[2 * i], gv = gtruth[2 * i + 1];
        if (unknown_flow(gu, gv)) continue;
        ++count;
        const float du = gu - flow[2 * i], dv = gv - flow[2 * i + 1];
        const float sq = du * du + dv * dv;     // float arithmetic as the reference's expression
        error += std::sqrt(sq);
    }
    return error / (double)count;
}

// ---- Flow::MotionToColor / computeColor / makecolorwheel (rw_flow.cpp:202-307) -------------
// The Middlebury colour wheel: 55 hues in six transitions whose lengths follow perceptual
// similarity.  Entry k of a transition of length n from colour A towards B moves one channel
// by 255*i/n (integer division), exactly as the reference's six loops do.
namespace {
struct ColorWheel {
    static constexpr int kCols = 15 + 6 + 4 + 11 + 13 + 6;
    int rgb[kCols][3];
    ColorWheel()
    {
        static const int len[6] = {15, 6, 4, 11, 13, 6};          // RY YG GC CB BM MR
        static const int moving[6] = {1, 0, 2, 1, 0, 2};          // channel that changes
        static const int rising[6] = {1, 0, 1, 0, 1, 0};          // ... upwards or downwards
        static const int base[6][3] = {{255, 0, 0}, {255, 255, 0}, {0, 255, 0}, {0, 255, 255}, {0, 0, 255}, {255, 0, 255}};
        int k = 0;
        for (int t = 0; t < 6; ++t)
            for (int i = 0; i < len[t]; ++i, ++k) {
                for (int c = 0; c < 3; ++c) rgb[k][c] = base[t][c];
                const int step = 255 * i / len[t];
                rgb[k][moving[t]] = rising[t] ? step : 255 - step;
            }
    }
};
const ColorWheel g_wheel;

// computeColor (rw_flow.cpp:251-275): hue from the angle, saturation from the radius; all float
// except the division by pi and the final scaling, which the reference's expressions do in double.
inline void compute_color(float fx, float fy, uint8_t *bgr)
{
    const int ncols = ColorWheel::kCols;
    const float rad = std::sqrt(fx * fx + fy * fy);
    const float a = (float)(std::atan2(-fy, -fx) / 3.14159265358979323846);   // M_PI
    const float fk = (a + 1.0f) / 2.0f * (float)(ncols - 1);
    const int k0 = (int)fk;
    const int k1 = (k0 + 1) % ncols;
    const float f = fk - (float)k0;
    for (int b = 0; b < 3; ++b) {
        const float col0 = (float)g_wheel.rgb[k0][b] / 255.0f;
        const float col1 = (float)g_wheel.rgb[k1][b] / 255.0f;
        float col = (1 - f) * col0 + f * col1;
        if (rad <= 1) col = 1 - rad * (1 - col);               // increase saturation with radius
        else col = (float)(col * .75);                          // out of range
        bgr[2 - b] = (uint8_t)(int)(255.0 * col);
    }
}
}  // namespace

void motion_to_color(const float *flow, int width, int height, float maxmotion, uint8_t *bgr, float range[5])
{
    float maxx = -999, maxy = -999, minx = 999, miny = 999, maxrad = -1;
    const size_t n = (size_t)width * height;
    for (size_t i = 0; i < n; ++i) {
        const float fx = flow[2 * i], fy = flow[2 * i + 1];
        if (unknown_flow(fx, fy)) continue;
        maxx = maxx > fx ? maxx : fx;
        maxy = maxy > fy ? maxy : fy;
        minx = minx < fx ? minx : fx;
        miny = miny < fy ? miny : fy;
        const float rad = std::sqrt(fx * fx + fy * fy);
        maxrad = maxrad > rad ? maxrad : rad;
    }
    if (range) { range[0] = maxrad; range[1] = minx; range[2] = maxx; range[3] = miny; range[4] = maxy; }
    if (maxmotion > 0) maxrad = maxmotion;                      // i.e. specified by the caller
    if (maxrad == 0) maxrad = 1;                                // flow == 0 everywhere
    for (size_t i = 0; i < n; ++i) {
        const float fx = flow[2 * i], fy = flow[2 * i + 1];
        uint8_t *pix = bgr + 3 * i;
        if (unknown_flow(fx, fy)) pix[0] = pix[1] = pix[2] = 0;
        else compute_color(fx / maxrad, fy / maxrad, pix);
    }
}

int ppm_write_bgr(const char *filename, int width, int height, const uint8_t *bgr)
{
    FILE *f = fopen(filename, "wb");
    if (!f) return fail(BBME_ERR_IO, "ppm_write: could not open %s", filename);
    fprintf(f, "P6\n%d %d\n255\n", width, height);
    std::vector<uint8_t> row((size_t)width * 3);
    bool ok = true;
    for (int y = 0; y < height && ok; ++y) {
        const uint8_t *s = bgr + (size_t)y * width * 3;
        for (int x = 0; x < width; ++x) { row[3 * x] = s[3 * x + 2]; row[3 * x + 1] = s[3 * x + 1]; row[3 * x + 2] = s[3 * x]; }
        ok = fwrite(row.data(), 1, row.size(), f) == row.size();
    }
    if (fclose(f) != 0) ok = false;
    return ok ? BBME_OK : fail(BBME_ERR_IO, "ppm_write: problem writing %s", filename);
}

void subsample_div4(const float *flow_padded, int padded_width, int padded_height,
                    int pad_x, int pad_y, float *out, int out_width)
{
    for (int i = pad_y; i < padded_height - pad_y; i += 4)
        for (int j = pad_x; j < padded_width - pad_x; j += 4) {
            const float *s = flow_padded + 2 * ((size_t)i * padded_width + j);
            float *d = out + 2 * ((size_t)((i - pad_y) / 4) * out_width + (j - pad_x) / 4);
            d[0] = s[0] / 4;
            d[1] = s[1] / 4;
        }
}

}  // namespace bbme

// ---------------------------------------------------------------------------------------
// C-ABI, host-only entry points
// ---------------------------------------------------------------------------------------
extern "C" {

const char *bbme_version(void) { return "bbme 0.1 (gfx950)"; }
const char *bbme_last_error(void) { return bbme::g_last_error.c_str(); }

int bbme_plan_padding(int width, int height, const bbme_params *params,
                      int *padded_width, int *padded_height, int *pad_x, int *pad_y)
{
    if (!params) return bbme::fail(BBME_ERR_INVALID, "params is null");
    if (int rc = bbme::validate_params(*params)) return rc;
    bbme::Geometry g;
    if (int rc = bbme::plan_padding(width, height, *params, g)) return rc;
    if (padded_width) *padded_width = g.padded_width;
    if (padded_height) *padded_height = g.padded_height;
    if (pad_x) *pad_x = g.pad_x;
    if (pad_y) *pad_y = g.pad_y;
    return BBME_OK;
}

int bbme_pad_zero_host(const uint8_t *src, int width, int height, int pitch, int pad_x, int pad_y, uint8_t *dst)
{
    if (!src || !dst || width <= 0 || height <= 0 || pitch < width || pad_x < 0 || pad_y < 0)
        return bbme::fail(BBME_ERR_INVALID, "bbme_pad_zero_host: bad arguments");
    bbme::pad_zero(src, width, height, pitch, pad_x, pad_y, dst);
    return BBME_OK;
}

int bbme_pyr_down_host(const uint8_t *src, int sw, int sh, uint8_t *dst)
{
    if (!src || !dst || sw < 2 || sh < 2) return bbme::fail(BBME_ERR_INVALID, "bbme_pyr_down_host: bad arguments");
    bbme::pyr_down(src, sw, sh, dst);
    return BBME_OK;
}

int bbme_resize_x4_host(const uint8_t *src, int sw, int sh, uint8_t *dst)
{
    if (!src || !dst || sw < 1 || sh < 1) return bbme::fail(BBME_ERR_INVALID, "bbme_resize_x4_host: bad arguments");
    bbme::resize_x4(src, sw, sh, dst);
    return BBME_OK;
}

int bbme_flo_read(const char *filename, int *width, int *height, float **data)
{
    if (!width || !height || !data) return bbme::fail(BBME_ERR_INVALID, "bbme_flo_read: null output");
    return bbme::flo_read(filename, width, height, data);
}

int bbme_flo_write(const char *filename, int width, int height, const float *data)
{
    return bbme::flo_write(filename, width, height, data);
}

int bbme_calculate_mse(const float *gtruth, const float *flow, int width, int height, double *out)
{
    if (!gtruth || !flow || !out || width < 1 || height < 1)
        return bbme::fail(BBME_ERR_INVALID, "bbme_calculate_mse: bad arguments");
    *out = bbme::calculate_mse(gtruth, flow, width, height);
    return BBME_OK;
}

int bbme_motion_to_color(const float *flow, int width, int height, float maxmotion, uint8_t *bgr, float *range)
{
    if (!flow || !bgr || width < 1 || height < 1)
        return bbme::fail(BBME_ERR_INVALID, "bbme_motion_to_color: bad arguments");
    bbme::motion_to_color(flow, width, height, maxmotion, bgr, range);
    return BBME_OK;
}

int bbme_ppm_write_bgr(const char *filename, int width, int height, const uint8_t *bgr)
{
    if (!filename || !bgr || width < 1 || height < 1)
        return bbme::fail(BBME_ERR_INVALID, "bbme_ppm_write_bgr: bad arguments");
    return bbme::ppm_write_bgr(filename, width, height, bgr);
}

int bbme_subsample_div4(const float *flow_padded, int padded_width, int padded_height,
                        int pad_x, int pad_y, float *out, int out_width, int out_height)
{
    if (!flow_padded || !out) return bbme::fail(BBME_ERR_INVALID, "bbme_subsample_div4: null pointer");
    const int need_w = (padded_width - 2 * pad_x + 3) / 4, need_h = (padded_height - 2 * pad_y + 3) / 4;
    if (out_width < need_w || out_height < need_h)
        return bbme::fail(BBME_ERR_INVALID, "bbme_subsample_div4: output %dx%d smaller than %dx%d",
                          out_width, out_height, need_w, need_h);
    bbme::subsample_div4(flow_padded, padded_width, padded_height, pad_x, pad_y, out, out_width);
    return BBME_OK;
}

int bbme_spiral_host(int search_size, int block_size, int16_t *dx, int16_t *dy, int capacity, int *count)
{
    if (!count) return bbme::fail(BBME_ERR_INVALID, "bbme_spiral_host: null count");
    const bbme::SpiralTable t = bbme::build_spiral(search_size, block_size);
    *count = (int)t.dx.size();
    if (dx && dy)
        for (int i = 0; i < *count && i < capacity; ++i) { dx[i] = t.dx[i]; dy[i] = t.dy[i]; }
    return BBME_OK;
}

int bbme_search_plan_host_waves(int range, int block_size, int waves, uint32_t *rounds, int rounds_capacity, int *nrounds,
                                uint32_t *tasks, int *groups, int *pitch_dw)
{
    if (!nrounds || range < 0 || range > 63 || (block_size != 8 && block_size != 16 && block_size != 32) || waves < 1 || waves > 2)
        return bbme::fail(BBME_ERR_INVALID, "bbme_search_plan_host: bad arguments");
    const int lanes = 64 * waves;
    const bbme::SearchPlan p = bbme::plan_search(range, block_size, waves == 2 || block_size == 32 ? 8 : 16, lanes);
    *nrounds = (int)p.rounds.size();
    if (groups) *groups = p.groups;
    if (pitch_dw) *pitch_dw = p.pitch_dw;
    for (int i = 0; i < *nrounds && i < rounds_capacity; ++i) {
        if (rounds) rounds[i] = p.rounds[i];
        if (tasks) memcpy(tasks + (size_t)i * lanes, p.tasks.data() + (size_t)i * lanes, lanes * sizeof(uint32_t));
    }
    return BBME_OK;
}

int bbme_search_plan_host(int range, int block_size, uint32_t *rounds, int rounds_capacity, int *nrounds,
                          uint32_t *tasks, int *groups, int *pitch_dw)
{
    return bbme_search_plan_host_waves(range, block_size, 1, rounds, rounds_capacity, nrounds, tasks, groups, pitch_dw);
}

void bbme_free(void *p) { free(p); }

// ---- asynchronous .flo writer: Flow::WriteFlowFile (rw_flow.cpp:139-200) on a worker thread ----
// The caller hands over a window of a (pinned) host buffer -- typically the padded field bbme_get_flow_host filled,
// with the padding stripped as main_class.cpp:63-70 does -- and goes on with the next pair; the worker writes the
// same bytes WriteFlowFile would ("PIEH", width, height, rows of interleaved u, v).  The buffer must stay untouched
// until bbme_flo_writer_wait returns.
struct bbme_flo_writer {
    struct Job {
        std::string name; int width, height; const float *data; size_t pitch_floats;
        const int16_t *cells; int cell_pitch, pad_x, pad_y;          // cells != nullptr: expand the 2x2-cell grid while writing
        unsigned long long ticket;
    };
    std::mutex mu;
    std::condition_variable cv_job, cv_idle;
    std::deque<Job> jobs;
    bool stop = false;
    int status = BBME_OK;
    std::string error;
    // tickets: job n is the n-th submitted (from 1); `finished` = every job up to it is on disk, `late` = finished jobs beyond
    // a gap (a pool of workers finishes files out of order)
    unsigned long long submitted = 0, finished = 0;
    std::set<unsigned long long> late;
    std::vector<std::thread> workers;
};

// copy_to_all_pixels (motion_framework.cpp:815-826) + the padding strip of main_class.cpp:63-70 + WriteFlowFile's rows, fused:
// pixel (x, y) of the file is the (dx, dy) of cell ((y + pad_y) / 2, (x + pad_x) / 2) as two floats.  The file is cut into bands
// of rows; a few threads expand bands into private buffers and pwrite them at their offsets (one thread fills ~5 GB/s, the
// 66 MB of a 4K field want more).
static bool flo_write_cells(const bbme_flo_writer::Job &job)
{
    const int fd = open(job.name.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return false;
    const size_t row_bytes = (size_t)job.width * 8;
    struct { char tag[4]; int32_t w, h; } head;
    memcpy(head.tag, bbme::kTagString, 4);
    head.w = job.width; head.h = job.height;
    // (measured on the GPU box's host, 66 MB per 4K field: 10 ms to tmpfs, 6 ms to the page cache, whatever the number of
    // threads -- the kernel's write path is the bound; mapping the file and expanding into the mapping was slower: 14-30 ms)
    std::atomic<bool> ok(pwrite(fd, &head, 12, 0) == 12);
    const int band = std::max(2, (int)((1u << 20) / row_bytes) & ~1);             // ~1 MB of output per band
    const int nbands = (job.height + band - 1) / band;
    int nthreads = 2;
    if (const char *e = getenv("BBME_WRITER_THREADS")) nthreads = atoi(e);
    nthreads = std::max(1, std::min(nthreads, std::min(nbands, 64)));
    std::atomic<int> next(0);
    auto work = [&]() {
        std::vector<float> buf((size_t)band * job.width * 2);
        for (int b = next.fetch_add(1); b < nbands && ok.load(); b = next.fetch_add(1)) {
            const int y0 = b * band, y1 = std::min(job.height, y0 + band);
            for (int y = y0; y < y1; ++y) {
                float *out = buf.data() + (size_t)(y - y0) * job.width * 2;
                const int py = y + job.pad_y;
                if (y > y0 && (py & 1)) {                                          // second row of a cell row: same values
                    memcpy(out, out - (size_t)job.width * 2, row_bytes);
                    continue;
                }
                const int16_t *crow = job.cells + (size_t)(py >> 1) * job.cell_pitch * 2;
                for (int x = 0; x < job.width; ++x) {
                    const int16_t *c = crow + (size_t)((x + job.pad_x) >> 1) * 2;
                    out[2 * x] = (float)c[0];
                    out[2 * x + 1] = (float)c[1];
                }
            }
            const size_t bytes = (size_t)(y1 - y0) * row_bytes;
            const char *src = reinterpret_cast<const char *>(buf.data());
            size_t done = 0;
            while (done < bytes) {
                const ssize_t n = pwrite(fd, src + done, bytes - done, (off_t)(12 + (size_t)y0 * row_bytes + done));
                if (n <= 0) { ok.store(false); break; }
                done += (size_t)n;
            }
        }
    };
    std::vector<std::thread> helpers;
    for (int i = 1; i < nthreads; ++i) helpers.emplace_back(work);
    work();
    for (auto &t : helpers) t.join();
    return (close(fd) == 0) && ok.load();
}

static void flo_writer_main(bbme_flo_writer *w)
{
    std::unique_lock<std::mutex> lk(w->mu);
    for (;;) {
        w->cv_job.wait(lk, [&] { return w->stop || !w->jobs.empty(); });
        if (w->jobs.empty()) return;
        bbme_flo_writer::Job job = w->jobs.front();
        w->jobs.pop_front();
        lk.unlock();
        bool ok = false;
        if (job.cells) ok = flo_write_cells(job);
        else if (FILE *f = fopen(job.name.c_str(), "wb")) {
            ok = fwrite(bbme::kTagString, 1, 4, f) == 4 && fwrite(&job.width, 4, 1, f) == 1 && fwrite(&job.height, 4, 1, f) == 1;
            const size_t row = (size_t)job.width * 2;
            if (job.pitch_floats == row) ok = ok && fwrite(job.data, sizeof(float), row * job.height, f) == row * job.height;
            else
                for (int y = 0; ok && y < job.height; ++y)
                    ok = fwrite(job.data + (size_t)y * job.pitch_floats, sizeof(float), row, f) == row;
            ok = (fclose(f) == 0) && ok;
        }
        lk.lock();
        if (!ok && w->status == BBME_OK) { w->status = BBME_ERR_IO; w->error = "WriteFlowFile: problem writing " + job.name; }
        w->late.insert(job.ticket);
        while (!w->late.empty() && *w->late.begin() == w->finished + 1) { w->late.erase(w->late.begin()); ++w->finished; }
        w->cv_idle.notify_all();
    }
}

int bbme_flo_writer_create(bbme_flo_writer **out) { return bbme_flo_writer_create_pool(1, out); }

int bbme_flo_writer_create_pool(int workers, bbme_flo_writer **out)
{
    if (!out) return bbme::fail(BBME_ERR_INVALID, "bbme_flo_writer_create: null output");
    if (workers < 1 || workers > 64) return bbme::fail(BBME_ERR_INVALID, "bbme_flo_writer_create_pool: %d workers (1..64)", workers);
    bbme_flo_writer *w = new bbme_flo_writer();
    for (int i = 0; i < workers; ++i) w->workers.emplace_back(flo_writer_main, w);
    *out = w;
    return BBME_OK;
}

int bbme_flo_writer_submit(bbme_flo_writer *w, const char *filename, int width, int height, const float *data,
                           int pitch_pixels)
{
    if (!w || !data || width < 1 || height < 1 || pitch_pixels < width)
        return bbme::fail(BBME_ERR_INVALID, "bbme_flo_writer_submit: bad arguments");
    if (!filename) return bbme::fail(BBME_ERR_IO, "WriteFlowFile: empty filename");
    const char *dot = strrchr(filename, '.');
    if (!dot) return bbme::fail(BBME_ERR_IO, "WriteFlowFile: extension required in filename");
    if (strcmp(dot, ".flo") != 0) return bbme::fail(BBME_ERR_IO, "WriteFlowFile: filename should have extension '.flo'");
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->jobs.push_back({filename, width, height, data, (size_t)pitch_pixels * 2, nullptr, 0, 0, 0, ++w->submitted});
    }
    w->cv_job.notify_one();
    return BBME_OK;
}

int bbme_flo_writer_submit_cells(bbme_flo_writer *w, const char *filename, int width, int height, const int16_t *cells,
                                 int cell_rows, int cell_cols, int pad_x, int pad_y)
{
    if (!w || !cells || width < 1 || height < 1 || pad_x < 0 || pad_y < 0 || cell_rows < 1 || cell_cols < 1)
        return bbme::fail(BBME_ERR_INVALID, "bbme_flo_writer_submit_cells: bad arguments");
    if (width + pad_x > 2 * cell_cols || height + pad_y > 2 * cell_rows)
        return bbme::fail(BBME_ERR_INVALID, "bbme_flo_writer_submit_cells: a %dx%d window at (%d, %d) does not fit %dx%d cells of 2x2",
                          width, height, pad_x, pad_y, cell_cols, cell_rows);
    if (!filename) return bbme::fail(BBME_ERR_IO, "WriteFlowFile: empty filename");
    const char *dot = strrchr(filename, '.');
    if (!dot) return bbme::fail(BBME_ERR_IO, "WriteFlowFile: extension required in filename");
    if (strcmp(dot, ".flo") != 0) return bbme::fail(BBME_ERR_IO, "WriteFlowFile: filename should have extension '.flo'");
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->jobs.push_back({filename, width, height, nullptr, 0, cells, cell_cols, pad_x, pad_y, ++w->submitted});
    }
    w->cv_job.notify_one();
    return BBME_OK;
}

int bbme_flo_writer_ticket(bbme_flo_writer *w, unsigned long long *ticket)
{
    if (!w || !ticket) return bbme::fail(BBME_ERR_INVALID, "bbme_flo_writer_ticket: null argument");
    std::lock_guard<std::mutex> lk(w->mu);
    *ticket = w->submitted;
    return BBME_OK;
}

int bbme_flo_writer_wait(bbme_flo_writer *w) { return bbme_flo_writer_wait_ticket(w, ~0ull); }

int bbme_flo_writer_wait_ticket(bbme_flo_writer *w, unsigned long long ticket)
{
    if (!w) return bbme::fail(BBME_ERR_INVALID, "bbme_flo_writer_wait: null writer");
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv_idle.wait(lk, [&] { return w->finished >= std::min(ticket, w->submitted); });
    if (w->status != BBME_OK) {
        const int st = w->status;
        const std::string msg = w->error;
        w->status = BBME_OK; w->error.clear();
        return bbme::fail(st, "%s", msg.c_str());
    }
    return BBME_OK;
}

int bbme_flo_writer_destroy(bbme_flo_writer *w)
{
    if (!w) return BBME_OK;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->stop = true;
    }
    w->cv_job.notify_all();
    for (std::thread &t : w->workers) t.join();      // they finish the queued files first
    delete w;
    return BBME_OK;
}

}  // extern "C"
