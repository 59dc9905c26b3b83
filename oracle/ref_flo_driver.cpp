// ref_flo_driver.cpp -- caller of the reference's OWN vendored Middlebury .flo code
// (middlebury/flow-code/flowIO.cpp: ReadFlowFile :46-92, WriteFlowFile :95-133).
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it only calls the
// reference's functions, which oracle/Makefile compiles from /root/reference in place
// into oracle/_ref/.  Used to pin orc_flo_read / orc_flo_write / bbme_flo_* .
//
//   flo_ref roundtrip <in.flo> <out.flo>   read with the reference, write with the reference
//   flo_ref stats <in.flo>                 print "w h n_unknown sum_u sum_v" of known pixels
//   flo_ref ramp <w> <h> <out.flo>         write a deterministic ramp field with the reference
//   flo_ref color <in.flo> <out.bgr> [maxmotion]   colour-code with the reference's vendored
//                                          computeColor (middlebury/flow-code/colorcode.cpp:52-77);
//                                          the normalisation around it is this driver's (after
//                                          color_flow.cpp:17-66); raw B,G,R bytes, no header
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "imageLib.h"
#include "flowIO.h"
#include "colorcode.h"
#include <cmath>

int main(int argc, char **argv)
{
    try {
        if (argc == 4 && !strcmp(argv[1], "roundtrip")) {
            CFloatImage img;
            ReadFlowFile(img, argv[2]);
            WriteFlowFile(img, argv[3]);
            return 0;
        }
        if (argc == 3 && !strcmp(argv[1], "stats")) {
            CFloatImage img;
            ReadFlowFile(img, argv[2]);
            CShape sh = img.Shape();
            long unknown = 0;
            double su = 0, sv = 0;
            for (int y = 0; y < sh.height; ++y)
                for (int x = 0; x < sh.width; ++x) {
                    float u = img.Pixel(x, y, 0), v = img.Pixel(x, y, 1);
                    if (unknown_flow(u, v)) { ++unknown; continue; }
                    su += u; sv += v;
                }
            printf("%d %d %ld %.17g %.17g\n", sh.width, sh.height, unknown, su, sv);
            return 0;
        }
        if (argc == 5 && !strcmp(argv[1], "ramp")) {
            int w = atoi(argv[2]), h = atoi(argv[3]);
            CShape sh(w, h, 2);
            CFloatImage img(sh);
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) {
                    img.Pixel(x, y, 0) = (float)(x - 3 * y) * 0.25f;
                    img.Pixel(x, y, 1) = (float)(7 * y - x) * 0.5f;
                }
            WriteFlowFile(img, argv[4]);
            return 0;
        }
        if ((argc == 4 || argc == 5) && !strcmp(argv[1], "color")) {
            CFloatImage img;
            ReadFlowFile(img, argv[2]);
            CShape sh = img.Shape();
            float maxrad = -1;
            for (int y = 0; y < sh.height; ++y)
                for (int x = 0; x < sh.width; ++x) {
                    float u = img.Pixel(x, y, 0), v = img.Pixel(x, y, 1);
                    if (unknown_flow(u, v)) continue;
                    float rad = std::sqrt(u * u + v * v);
                    if (rad > maxrad) maxrad = rad;
                }
            if (argc == 5 && atof(argv[4]) > 0) maxrad = (float)atof(argv[4]);
            if (maxrad == 0) maxrad = 1;
            FILE *f = fopen(argv[3], "wb");
            if (!f) return 2;
            for (int y = 0; y < sh.height; ++y)
                for (int x = 0; x < sh.width; ++x) {
                    float u = img.Pixel(x, y, 0), v = img.Pixel(x, y, 1);
                    uchar pix[3] = {0, 0, 0};
                    if (!unknown_flow(u, v)) computeColor(u / maxrad, v / maxrad, pix);
                    fwrite(pix, 1, 3, f);
                }
            fclose(f);
            return 0;
        }
    } catch (CError &err) {
        fprintf(stderr, "flo_ref: %s\n", err.message);
        return 2;
    }
    fprintf(stderr, "usage: flo_ref roundtrip in out | stats in | ramp w h out | color in out [maxmotion]\n");
    return 1;
}
