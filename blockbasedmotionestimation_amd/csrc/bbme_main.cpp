// bbme_main.cpp -- the reference's driver (main_class.cpp:6-85) as a real command line.
//
//   bbme_cli frame10.pgm frame11.pgm [--gt flow10.flo] [--out flow.flo] [--color flow.ppm] [--levels N]
//            [--block B] [--search S] [--no-upsample] [--device D]
//
// Sequence of main_class.cpp: read two grey frames (:24,26; binary PGM here, the image has no
// libpng), 4x bilinear up-sampling (:32-33), MF::MF (:45), timed calcMotionBlockMatching (:47-55),
// strip the padding + every 4th pixel / 4 (:58-70), write the field (the reference only ever
// colour-codes it; here Flow::WriteFlowFile is actually called), EPE against ground truth (:78-82).
// Defaults are the reference's literals (:19-21): 4 levels, block 32, search 64.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "rw_flow.hpp"

static bool read_pgm(const char *path, bbme::Image8 &img)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    char magic[3] = {0, 0, 0};
    int w = 0, h = 0, maxv = 0;
    auto skip = [&]() {
        int c;
        while ((c = fgetc(f)) != EOF) {
            if (c == '#') { while ((c = fgetc(f)) != EOF && c != '\n') {} }
            else if (c != ' ' && c != '\n' && c != '\r' && c != '\t') { ungetc(c, f); break; }
        }
    };
    bool ok = fread(magic, 1, 2, f) == 2 && magic[0] == 'P' && magic[1] == '5';
    if (ok) { skip(); ok = fscanf(f, "%d", &w) == 1; }
    if (ok) { skip(); ok = fscanf(f, "%d", &h) == 1; }
    if (ok) { skip(); ok = fscanf(f, "%d", &maxv) == 1 && maxv == 255; }
    if (ok) ok = fgetc(f) != EOF && w > 0 && h > 0;
    if (ok) {
        img = bbme::Image8(h, w);
        ok = fread(img.data.data(), 1, img.data.size(), f) == img.data.size();
    }
    fclose(f);
    return ok;
}

int main(int argc, char **argv)
{
    const char *f1 = nullptr, *f2 = nullptr, *gt = nullptr, *out = nullptr, *color = nullptr;
    int levels = 4, block = 32, search = 64, device = 0;
    bool upsample = true;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--gt") gt = next();
        else if (a == "--out") out = next();
        else if (a == "--color") color = next();
        else if (a == "--levels") levels = atoi(next());
        else if (a == "--block") block = atoi(next());
        else if (a == "--search") search = atoi(next());
        else if (a == "--device") device = atoi(next());
        else if (a == "--no-upsample") upsample = false;
        else if (!f1) f1 = argv[i];
        else if (!f2) f2 = argv[i];
        else { fprintf(stderr, "unexpected argument %s\n", argv[i]); return 2; }
    }
    if (!f1 || !f2 || levels < 1 || levels > BBME_MAX_LEVELS) {
        fprintf(stderr, "usage: bbme_cli frame1.pgm frame2.pgm [--gt gt.flo] [--out flow.flo] [--color flow.ppm] "
                        "[--levels N] [--block B] [--search S] [--no-upsample] [--device D]\n");
        return 2;
    }
    try {
        bbme::Image8 image1, image2;
        if (!read_pgm(f1, image1) || !read_pgm(f2, image2)) {
            fprintf(stderr, "Could not open one of the images\n");                    // main_class.cpp:40
            return 1;
        }
        const int orig_height = image1.rows, orig_width = image1.cols;
        const int scale = upsample ? 4 : 1;
        if (upsample) { image1 = bbme::resize_x4(image1); image2 = bbme::resize_x4(image2); }
        std::vector<int> search_size(levels, search), block_size(levels, block);
        MF motion_pair(image1, image2, search_size.data(), block_size.data(), levels, device);
        const auto t1 = std::chrono::steady_clock::now();
        bbme::ImageFlow flow_res = motion_pair.calcMotionBlockMatching();
        const auto t2 = std::chrono::steady_clock::now();
        printf("Seconds: %g\n", std::chrono::duration<double>(t2 - t1).count());
        bbme::ImageFlow subpix(orig_height, orig_width);
        if (upsample) {
            bbme::check(bbme_subsample_div4(flow_res.data.data(), motion_pair.padded_width, motion_pair.padded_height,
                                            motion_pair.padding_x, motion_pair.padding_y, subpix.data.data(),
                                            orig_width, orig_height));
        } else {
            for (int y = 0; y < orig_height; ++y)
                for (int x = 0; x < orig_width; ++x) {
                    const float *s = flow_res.at(y + motion_pair.padding_y, x + motion_pair.padding_x);
                    subpix.at(y, x)[0] = s[0];
                    subpix.at(y, x)[1] = s[1];
                }
        }
        (void)scale;
        Flow file;
        if (color) {                                   // main_class.cpp:73-75 (flow.png there)
            bbme::ImageBGR flow_img;
            file.MotionToColor(subpix, flow_img, -1);
            file.ShowImage(flow_img, color);
        }
        if (out) file.WriteFlowFile(subpix, out);
        if (gt) {
            bbme::ImageFlow gtruth;
            file.ReadFlowFile(gtruth, gt);
            printf("Calculated MSE is %.9g\n", file.CalculateMSE(gtruth, subpix));       // :82
        }
    } catch (const bbme::Error &e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
