// bbme_seq_main.cpp -- a sequence of frame pairs over the GPUs of one node, C++ only (no torch).
//
//   bbme_seq --gpus N [--levels L] [--block B] [--search S] --out DIR  f0a.pgm f0b.pgm  f1a.pgm f1b.pgm ...
//
// Pair p runs on GPU p mod N (one context per GPU, whole pyramid, no exchange: SURVEY.md 8e); after every round of N
// pairs the compact cell grids are gathered on GPU 0 with one ncclGather (bbme_gather_cells), downloaded in one copy and
// written as DIR/0000.flo, 0001.flo, ... by the asynchronous writer, which expands them to the dense fields as it
// writes, while the next round runs.
// One process drives all N GPUs here (ncclCommInitAll); bbme_gather_cells itself does not care who owns the ranks.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bbme_rccl.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define NCCL_OK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 1; } } while (0)
#define BBME_OKAY(x) do { int s_ = (x); if (s_ != BBME_OK) { fprintf(stderr, "%s: %s\n", #x, bbme_last_error()); return 1; } } while (0)

static bool read_pgm(const char *path, std::vector<uint8_t> &px, int &w, int &h)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    int maxv = 0;
    bool ok = fgetc(f) == 'P' && fgetc(f) == '5';
    auto skip = [&]() {
        int c;
        while ((c = fgetc(f)) != EOF) {
            if (c == '#') { while ((c = fgetc(f)) != EOF && c != '\n') {} }
            else if (c != ' ' && c != '\n' && c != '\r' && c != '\t') { ungetc(c, f); break; }
        }
    };
    if (ok) { skip(); ok = fscanf(f, "%d", &w) == 1; }
    if (ok) { skip(); ok = fscanf(f, "%d", &h) == 1; }
    if (ok) { skip(); ok = fscanf(f, "%d", &maxv) == 1 && maxv == 255; }
    if (ok) ok = fgetc(f) != EOF && w > 0 && h > 0;
    if (ok) { px.resize((size_t)w * h); ok = fread(px.data(), 1, px.size(), f) == px.size(); }
    fclose(f);
    return ok;
}

int main(int argc, char **argv)
{
    int gpus = 1, levels = 4, block = 16, search = 80;
    const char *out_dir = nullptr;
    std::vector<const char *> files;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--gpus") gpus = atoi(next());
        else if (a == "--levels") levels = atoi(next());
        else if (a == "--block") block = atoi(next());
        else if (a == "--search") search = atoi(next());
        else if (a == "--out") out_dir = next();
        else files.push_back(argv[i]);
    }
    if (gpus < 1 || files.empty() || files.size() % 2 || !out_dir || levels < 1 || levels > BBME_MAX_LEVELS) {
        fprintf(stderr, "usage: bbme_seq --gpus N [--levels L] [--block B] [--search S] --out DIR f0a.pgm f0b.pgm [f1a.pgm f1b.pgm ...]\n");
        return 2;
    }
    const int n_pairs = (int)files.size() / 2;
    int ndev = 0;
    HIP_OK(hipGetDeviceCount(&ndev));
    if (gpus > ndev) { fprintf(stderr, "--gpus %d but %d device(s) visible\n", gpus, ndev); return 1; }

    std::vector<std::vector<uint8_t>> img(files.size());
    int w = 0, h = 0;
    for (size_t i = 0; i < files.size(); ++i) {
        int wi = 0, hi = 0;
        if (!read_pgm(files[i], img[i], wi, hi)) { fprintf(stderr, "Could not open %s\n", files[i]); return 1; }
        if (i && (wi != w || hi != h)) { fprintf(stderr, "%s: all frames must have one size\n", files[i]); return 1; }
        w = wi; h = hi;
    }
    bbme_params params{};
    params.num_levels = levels;
    for (int l = 0; l < levels; ++l) { params.block_size[l] = block; params.search_size[l] = search; }

    std::vector<int> devs(gpus);
    for (int r = 0; r < gpus; ++r) devs[r] = r;
    std::vector<ncclComm_t> comms(gpus);
    NCCL_OK(ncclCommInitAll(comms.data(), gpus, devs.data()));
    std::vector<bbme_ctx *> ctx(gpus, nullptr);
    for (int r = 0; r < gpus; ++r) {
        BBME_OKAY(bbme_create(&params, w, h, r, &ctx[r]));
    }
    int pw = 0, ph = 0, pad_x = 0, pad_y = 0;
    BBME_OKAY(bbme_get_geometry(ctx[0], &pw, &ph, &pad_x, &pad_y));
    const size_t words = (size_t)(pw / 2) * (ph / 2);
    int32_t *d_recv = nullptr;
    HIP_OK(hipSetDevice(0));
    HIP_OK(hipMalloc(&d_recv, words * gpus * sizeof(int32_t)));
    void *stream0 = nullptr;
    BBME_OKAY(bbme_get_stream(ctx[0], &stream0));
    // pinned staging for the writer: the gathered cell grids of a round (1/16 of the dense fields), double-buffered over
    // rounds; the writer expands them while it writes (bbme_flo_writer_submit_cells)
    int32_t *host[2] = {nullptr, nullptr};
    for (auto &p : host) HIP_OK(hipHostMalloc(&p, words * gpus * sizeof(int32_t)));
    bbme_flo_writer *writer = nullptr;
    BBME_OKAY(bbme_flo_writer_create(&writer));

    const auto t0 = std::chrono::steady_clock::now();
    const int rounds = (n_pairs + gpus - 1) / gpus;
    for (int k = 0; k < rounds; ++k) {
        for (int r = 0; r < gpus; ++r) {
            const int p = k * gpus + r;
            if (p >= n_pairs) continue;                       // this rank idles in the last round but still joins the gather
            BBME_OKAY(bbme_set_frames_host(ctx[r], img[2 * p].data(), img[2 * p + 1].data(), w));
            BBME_OKAY(bbme_estimate(ctx[r]));
        }
        NCCL_OK(ncclGroupStart());
        for (int r = 0; r < gpus; ++r) BBME_OKAY(bbme_gather_cells(ctx[r], comms[r], 0, d_recv));
        NCCL_OK(ncclGroupEnd());
        if (k >= 2) BBME_OKAY(bbme_flo_writer_wait(writer));  // the staging buffers of round k - 2 are free again
        int32_t *dst = host[k & 1];
        HIP_OK(hipMemcpyAsync(dst, d_recv, words * gpus * sizeof(int32_t), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream0)));
        HIP_OK(hipStreamSynchronize(static_cast<hipStream_t>(stream0)));
        for (int r = 0; r < gpus; ++r) {
            const int p = k * gpus + r;
            if (p >= n_pairs) continue;
            char name[32];
            snprintf(name, sizeof name, "/%04d.flo", p);
            const std::string path = std::string(out_dir) + name;
            BBME_OKAY(bbme_flo_writer_submit_cells(writer, path.c_str(), w, h, reinterpret_cast<const int16_t *>(dst + (size_t)r * words),
                                                   ph / 2, pw / 2, pad_x, pad_y));
        }
    }
    for (int r = 0; r < gpus; ++r) BBME_OKAY(bbme_synchronize(ctx[r]));       // also refuses a field that did not converge
    BBME_OKAY(bbme_flo_writer_wait(writer));
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%d pairs of %dx%d on %d GPU(s): %.3f s (%.2f ms per pair, files included)\n", n_pairs, w, h, gpus, secs,
           secs / n_pairs * 1e3);

    bbme_flo_writer_destroy(writer);
    for (auto p : host) (void)hipHostFree(p);
    (void)hipFree(d_recv);
    for (int r = 0; r < gpus; ++r) { bbme_destroy(ctx[r]); ncclCommDestroy(comms[r]); }
    return 0;
}
