// motion_framework.hpp -- C++ host-side mirror of the reference's MF class (motion_framework.h:9-54)
// over the C-ABI of libbbme.so.  Same class name, constructor argument order and public fields, so
// code written against the reference's MF compiles against this one after swapping cv::Mat for
// bbme::Image (or with -DBBME_WITH_OPENCV for the cv::Mat overloads).  Header only.
//
//   reference                                              here
//   MF(cv::Mat&, cv::Mat&, const int[], const int[], int)  MF(const Image8&, const Image8&, const int[], const int[], int)
//   cv::Mat calcMotionBlockMatching()                      ImageFlow calcMotionBlockMatching()
//   padded_height / padded_width / padding_x / padding_y   same public ints
//
// Errors: the reference asserts or prints and exit(1)s (motion_framework.cpp:7-8,21-26); this
// class throws bbme::Error carrying the C-ABI status and message.  Unlike the reference's MF, an
// instance may run calcMotionBlockMatching() any number of times (it is not one-shot).
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "bbme.h"

#ifdef BBME_WITH_OPENCV
#include <opencv2/core/core.hpp>
#endif

namespace bbme {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string &m) : std::runtime_error("bbme status " + std::to_string(s) + ": " + m), status(s) {}
};

inline void check(int status)
{
    if (status != BBME_OK) throw Error(status, bbme_last_error());
}

// 8-bit single-channel image (the reference's CV_8UC1 cv::Mat), row-major, pitch == cols
struct Image8 {
    int rows = 0, cols = 0;
    std::vector<uint8_t> data;
    Image8() = default;
    Image8(int r, int c) : rows(r), cols(c), data((size_t)r * c) {}
    uint8_t &at(int y, int x) { return data[(size_t)y * cols + x]; }
    uint8_t at(int y, int x) const { return data[(size_t)y * cols + x]; }
};

// two-band float image (the reference's CV_32FC2 cv::Mat): (u, v) = (dx, dy) interleaved
struct ImageFlow {
    int rows = 0, cols = 0;
    std::vector<float> data;
    ImageFlow() = default;
    ImageFlow(int r, int c) : rows(r), cols(c), data((size_t)r * c * 2) {}
    float *at(int y, int x) { return &data[2 * ((size_t)y * cols + x)]; }
    const float *at(int y, int x) const { return &data[2 * ((size_t)y * cols + x)]; }
};

// three-band 8-bit image, B,G,R interleaved (the reference's CV_8UC3 cv::Mat)
struct ImageBGR {
    int rows = 0, cols = 0;
    std::vector<uint8_t> data;
    ImageBGR() = default;
    ImageBGR(int r, int c) : rows(r), cols(c), data((size_t)r * c * 3) {}
    uint8_t *at(int y, int x) { return &data[3 * ((size_t)y * cols + x)]; }
    const uint8_t *at(int y, int x) const { return &data[3 * ((size_t)y * cols + x)]; }
};

// cv::resize(img, img, cv::Size(), 4, 4, cv::INTER_LINEAR) of main_class.cpp:32-33
inline Image8 resize_x4(const Image8 &src)
{
    Image8 dst(src.rows * 4, src.cols * 4);
    check(bbme_resize_x4_host(src.data.data(), src.cols, src.rows, dst.data.data()));
    return dst;
}

}  // namespace bbme

class MF {
public:
    MF(const bbme::Image8 &image1, const bbme::Image8 &image2, const int search_size[], const int block_size[],
       const int num_levels, int device = 0)
    {
        if (num_levels <= 0) throw bbme::Error(BBME_ERR_INVALID, "num_levels must be > 0");                       // assert :7
        if (image1.rows != image2.rows || image1.cols != image2.cols)
            throw bbme::Error(BBME_ERR_INVALID, "image1.size() != image2.size()");                                 // assert :8
        bbme_params p{};
        p.num_levels = num_levels;
        for (int i = 0; i < num_levels && i < BBME_MAX_LEVELS; ++i) {
            p.block_size[i] = block_size[i];
            p.search_size[i] = search_size[i];
        }
        bbme::check(bbme_create(&p, image1.cols, image1.rows, device, &ctx_));
        bbme::check(bbme_get_geometry(ctx_, &padded_width, &padded_height, &padding_x, &padding_y));
        int rc = bbme_set_frames_host(ctx_, image1.data.data(), image2.data.data(), image1.cols);
        if (rc != BBME_OK) { bbme_destroy(ctx_); ctx_ = nullptr; bbme::check(rc); }
    }
#ifdef BBME_WITH_OPENCV
    MF(cv::Mat &image1, cv::Mat &image2, const int search_size[], const int block_size[], const int num_levels, int device = 0)
        : MF(from_mat(image1), from_mat(image2), search_size, block_size, num_levels, device) {}
#endif
    MF(const MF &) = delete;
    MF &operator=(const MF &) = delete;
    ~MF() { if (ctx_) bbme_destroy(ctx_); }

    // Perform block matching for the whole hierarchy/pyramid (motion_framework.cpp:113-219).
    // Returns the dense padded field (padded_height x padded_width, CV_32FC2 layout).
    bbme::ImageFlow calcMotionBlockMatching()
    {
        bbme::check(bbme_estimate(ctx_));
        bbme::ImageFlow flow(padded_height, padded_width);
        bbme::check(bbme_get_flow_host(ctx_, flow.data.data()));
        return flow;
    }
#ifdef BBME_WITH_OPENCV
    cv::Mat calcMotionBlockMatchingMat()
    {
        bbme::check(bbme_estimate(ctx_));
        cv::Mat flow(padded_height, padded_width, CV_32FC2);
        bbme::check(bbme_get_flow_host(ctx_, reinterpret_cast<float *>(flow.data)));
        return flow;
    }
#endif
    bbme_ctx *context() { return ctx_; }

    int padded_height = 0;        // motion_framework.h:16-19
    int padded_width = 0;
    int padding_x = 0;
    int padding_y = 0;

private:
#ifdef BBME_WITH_OPENCV
    static bbme::Image8 from_mat(const cv::Mat &m)
    {
        bbme::Image8 im(m.rows, m.cols);
        for (int y = 0; y < m.rows; ++y) std::copy(m.ptr<uint8_t>(y), m.ptr<uint8_t>(y) + m.cols, &im.data[(size_t)y * m.cols]);
        return im;
    }
#endif
    bbme_ctx *ctx_ = nullptr;
};
