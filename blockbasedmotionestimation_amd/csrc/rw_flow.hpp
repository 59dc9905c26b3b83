// rw_flow.hpp -- C++ host-side mirror of the reference's Flow class (rw_flow.h:9-38) over the C-ABI.
// ReadFlowFile / WriteFlowFile / MotionToColor / CalculateMSE keep the reference's names and
// argument order.  ShowImage (rw_flow.cpp:334-340: GUI window + flowimg.png) only writes the
// image, as binary PPM (no GUI, no PNG codec here).  Errors throw bbme::Error instead of exit(1).
#pragma once

#include "motion_framework.hpp"

class Flow {
public:
    // read a flow file into 2-band image (rw_flow.cpp:50-136)
    void ReadFlowFile(bbme::ImageFlow &img, const char *filename)
    {
        int w = 0, h = 0;
        float *data = nullptr;
        bbme::check(bbme_flo_read(filename, &w, &h, &data));
        img = bbme::ImageFlow(h, w);
        std::copy(data, data + (size_t)w * h * 2, img.data.begin());
        bbme_free(data);
    }
    // write a 2-band image into flow file (rw_flow.cpp:139-200)
    void WriteFlowFile(const bbme::ImageFlow &img, const char *filename)
    {
        bbme::check(bbme_flo_write(filename, img.cols, img.rows, img.data.data()));
    }
    // colour coding of a flow field (rw_flow.cpp:202-249); prints the reference's range line
    void MotionToColor(const bbme::ImageFlow &input_img, bbme::ImageBGR &output_img, float maxmotion)
    {
        output_img = bbme::ImageBGR(input_img.rows, input_img.cols);
        float r[5];
        bbme::check(bbme_motion_to_color(input_img.data.data(), input_img.cols, input_img.rows, maxmotion,
                                         output_img.data.data(), r));
        printf("max motion: %.4f  motion range: u = %.3f .. %.3f;  v = %.3f .. %.3f\n", r[0], r[1], r[2], r[3], r[4]);
    }
    void ShowImage(const bbme::ImageBGR &flow_img, const char *filename = "flowimg.ppm")
    {
        bbme::check(bbme_ppm_write_bgr(filename, flow_img.cols, flow_img.rows, flow_img.data.data()));
    }
    // "mean-squared error" of the reference = mean end-point error over known GT pixels (rw_flow.cpp:309-332)
    double CalculateMSE(const bbme::ImageFlow &gtruth, const bbme::ImageFlow &flow)
    {
        if (gtruth.rows != flow.rows || gtruth.cols != flow.cols)
            throw bbme::Error(BBME_ERR_INVALID, "CalculateMSE: sizes differ");
        double out = 0;
        bbme::check(bbme_calculate_mse(gtruth.data.data(), flow.data.data(), gtruth.cols, gtruth.rows, &out));
        return out;
    }
};
