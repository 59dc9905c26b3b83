// rw_flow.hpp -- C++ host-side mirror of the reference's Flow class (rw_flow.h:9-38) over the C-ABI.
// ReadFlowFile / WriteFlowFile / CalculateMSE keep the reference's names and argument order.
// MotionToColor / ShowImage (colour wheel + GUI window, rw_flow.cpp:202-307,334-340) are
// visualisation only and out of scope.  Errors throw bbme::Error instead of exit(1).
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
