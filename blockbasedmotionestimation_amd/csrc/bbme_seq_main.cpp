// bbme_seq_main.cpp -- a sequence of frame pairs over the GPUs of one node, C++ only (no torch).
//
//   bbme_seq --gpus N [--levels L] [--block B] [--search S] --out DIR  f0a.pgm f0b.pgm  f1a.pgm f1b.pgm ...
//
// Pair p runs on GPU p mod N (one context per GPU, whole pyramid, no exchange: SURVEY.md 8e); after every round of N
// pairs the compact cell grids are gathered on GPU 0 with one ncclGather (bbme_gather_cells), downloaded in one copy and
// written as DIR/0000.flo, 0001.flo, ... by a pool of asynchronous writers (--writers W, default 3: one file each at a
// time), which expand them to the dense fields as they write.  The round loop is csrc/seq_schedule.hpp (run_sequence):
// a reader thread keeps a ring of pinned frame slots (three rounds of pairs) filled ahead of the loop, frames go up
// without a host wait (bbme_set_frames_host_async), receive and staging buffers are double-buffered, the download of round
// k runs on a copy stream beside the estimates of round k + 1, and the host only ever waits for work at least a round old
// -- for the writers only until the files of round k - 2 are on disk, not for everything submitted.
// One process drives all N GPUs here (ncclCommInitAll); bbme_gather_cells itself does not care who owns the ranks.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bbme_rccl.h"
#include "seq_schedule.hpp"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define NCCL_OK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); exit(1); } } while (0)
#define BBME_OKAY(x) do { int s_ = (x); if (s_ != BBME_OK) { fprintf(stderr, "%s: %s\n", #x, bbme_last_error()); exit(1); } } while (0)

// binary PGM (P5, maxval 255): the header, then (`px` given) w * h bytes into px
static bool read_pgm(const char *path, uint8_t *px, int &w, int &h)
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
    int wi = 0, hi = 0;
    if (ok) { skip(); ok = fscanf(f, "%d", &wi) == 1; }
    if (ok) { skip(); ok = fscanf(f, "%d", &hi) == 1; }
    if (ok) { skip(); ok = fscanf(f, "%d", &maxv) == 1 && maxv == 255; }
    if (ok) ok = fgetc(f) != EOF && wi > 0 && hi > 0;
    if (ok && px) ok = wi == w && hi == h && fread(px, 1, (size_t)wi * hi, f) == (size_t)wi * hi;
    if (ok && !px) { w = wi; h = hi; }
    fclose(f);
    return ok;
}

namespace {

// The frames of the sequence, a few rounds at a time: `slots` pinned pair slots (pair p lives in slot p mod slots), filled in
// pair order by a reader thread as slots come free.  The memory is pinned once (hipHostMalloc), so an upload from a slot is
// asynchronous; at no time are more than `slots` pairs in memory, however long the sequence.
struct FrameRing {
    int slots = 0, n_pairs = 0, w = 0, h = 0;
    size_t bytes = 0;
    uint8_t *mem = nullptr;
    std::vector<const char *> files;
    std::vector<int> holds;                         // pair in the slot (loaded), -1 = free
    std::mutex mu;
    std::condition_variable cv;
    std::thread reader;
    std::string error;
    bool stop = false;

    void start(const std::vector<const char *> &f, int w_, int h_, int slots_)
    {
        files = f; w = w_; h = h_; slots = slots_; n_pairs = (int)f.size() / 2;
        bytes = ((size_t)w * h + 255) / 256 * 256;
        HIP_OK(hipHostMalloc(&mem, bytes * 2 * slots, hipHostMallocPortable));     // every GPU uploads from it
        holds.assign(slots, -1);
        reader = std::thread([this] {
            for (int p = 0; p < n_pairs; ++p) {
                const int s = p % slots;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || holds[s] < 0; });
                    if (stop) return;
                }
                bool ok = true;
                for (int i = 0; i < 2 && ok; ++i) {
                    int wi = w, hi = h;
                    ok = read_pgm(files[2 * p + i], mem + (size_t)(2 * s + i) * bytes, wi, hi);
                    if (!ok) { std::lock_guard<std::mutex> lk(mu); error = std::string("Could not read ") + files[2 * p + i] + " (binary PGM of the sequence's size)"; }
                }
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (!ok) stop = true;
                    else holds[s] = p;
                }
                cv.notify_all();
                if (!ok) return;
            }
        });
    }
    // the two frames of pair p, once the reader has them
    const uint8_t *frame(int p, int which)
    {
        const int s = p % slots;
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || holds[s] == p; });
        if (holds[s] != p) { fprintf(stderr, "%s\n", error.empty() ? "frame reader stopped" : error.c_str()); exit(1); }
        return mem + (size_t)(2 * s + which) * bytes;
    }
    void release(int p)
    {
        { std::lock_guard<std::mutex> lk(mu); if (holds[p % slots] == p) holds[p % slots] = -1; }
        cv.notify_all();
    }
    void finish()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        if (reader.joinable()) reader.join();
        if (mem) (void)hipHostFree(mem);
        mem = nullptr;
    }
};

// run_sequence's backend over HIP streams, RCCL and the asynchronous writers (see seq_schedule.hpp for the contract)
struct HipBackend {
    int gpus = 0, n_pairs = 0, w = 0, h = 0, pw = 0, ph = 0, pad_x = 0, pad_y = 0;
    size_t words = 0;
    FrameRing *frames = nullptr;
    std::vector<bbme_ctx *> ctx;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> stream;               // the contexts' streams
    hipStream_t copy_stream = nullptr;             // GPU 0: downloads beside the next round's estimates
    // receive (HBM) and staging (pinned host) buffers of a round's gathered cell grids: a ring of `nbuf`, deep enough to keep
    // every writer busy (a round's files stay in its staging buffer until they are written)
    int nbuf = 2;
    std::vector<int32_t *> d_recv, host;
    std::vector<hipEvent_t> ev_gathered, ev_downloaded;
    std::vector<char> downloaded_once;
    bbme_flo_writer *writer = nullptr;              // a pool: one file per worker at a time
    std::vector<unsigned long long> round_ticket;   // per round: the ticket of its last file
    std::string out_dir;
    // per-round phase stamps of GPU 0 (timing events): start, frames up + pyramid, estimate, gather on its stream; download on the copy stream
    std::vector<hipEvent_t> t_start, t_frames, t_estimate, t_gather, t_download;
    int round = 0;

    hipEvent_t stamp(hipStream_t s)
    {
        hipEvent_t e;
        HIP_OK(hipSetDevice(0));
        HIP_OK(hipEventCreate(&e));
        HIP_OK(hipEventRecord(e, s));
        return e;
    }
    void upload(int r, int pair)
    {
        if (r == 0) t_start.push_back(stamp(stream[0]));
        const uint8_t *f1 = frames->frame(pair, 0), *f2 = frames->frame(pair, 1);     // waits for the reader only if it has fallen behind
        BBME_OKAY(bbme_set_frames_host_async(ctx[r], 0, f1, f2, w));                  // no host wait
        if (r == 0) t_frames.push_back(stamp(stream[0]));
    }
    void estimate(int r)
    {
        BBME_OKAY(bbme_estimate(ctx[r]));
        if (r == 0) t_estimate.push_back(stamp(stream[0]));
    }
    void root_wait_downloaded(int b)
    {
        HIP_OK(hipSetDevice(0));
        if (downloaded_once[b]) HIP_OK(hipStreamWaitEvent(stream[0], ev_downloaded[b], 0));
    }
    void gather(int b)
    {
        NCCL_OK(ncclGroupStart());
        for (int r = 0; r < gpus; ++r) BBME_OKAY(bbme_gather_cells(ctx[r], comms[r], 0, d_recv[b]));
        NCCL_OK(ncclGroupEnd());
    }
    void record_gathered(int b)
    {
        HIP_OK(hipSetDevice(0));
        HIP_OK(hipEventRecord(ev_gathered[b], stream[0]));
        t_gather.push_back(stamp(stream[0]));
    }
    void host_wait_writer(int k)                     // the files of round k, not everything submitted
    {
        if (k >= 0 && k < (int)round_ticket.size()) BBME_OKAY(bbme_flo_writer_wait_ticket(writer, round_ticket[k]));
    }
    void release_frames(int k)
    {
        for (int r = 0; r < gpus; ++r)
            if (k * gpus + r < n_pairs) frames->release(k * gpus + r);
    }
    void download(int b)
    {
        HIP_OK(hipSetDevice(0));
        HIP_OK(hipStreamWaitEvent(copy_stream, ev_gathered[b], 0));
        HIP_OK(hipMemcpyAsync(host[b], d_recv[b], words * gpus * sizeof(int32_t), hipMemcpyDeviceToHost, copy_stream));
    }
    void record_downloaded(int b)
    {
        HIP_OK(hipSetDevice(0));
        HIP_OK(hipEventRecord(ev_downloaded[b], copy_stream));
        downloaded_once[b] = true;
        t_download.push_back(stamp(copy_stream));
    }
    void host_wait_downloaded(int b) { HIP_OK(hipEventSynchronize(ev_downloaded[b])); }
    void submit_files(int k, int b)
    {
        for (int r = 0; r < gpus; ++r) {
            const int p = k * gpus + r;
            if (p >= n_pairs) continue;
            char name[32];
            snprintf(name, sizeof name, "/%04d.flo", p);
            const std::string path = out_dir + name;
            BBME_OKAY(bbme_flo_writer_submit_cells(writer, path.c_str(), w, h, reinterpret_cast<const int16_t *>(host[b] + (size_t)r * words),
                                                   ph / 2, pw / 2, pad_x, pad_y));
        }
        if ((int)round_ticket.size() <= k) round_ticket.resize(k + 1, 0);
        BBME_OKAY(bbme_flo_writer_ticket(writer, &round_ticket[k]));
    }
};

}  // namespace

int main(int argc, char **argv)
{
    int gpus = 1, levels = 4, block = 16, search = 80, writers = 3;
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
        else if (a == "--writers") writers = atoi(next());
        else files.push_back(argv[i]);
    }
    if (gpus < 1 || files.empty() || files.size() % 2 || !out_dir || levels < 1 || levels > BBME_MAX_LEVELS || writers < 1 || writers > 64) {
        fprintf(stderr, "usage: bbme_seq --gpus N [--levels L] [--block B] [--search S] [--writers W] --out DIR f0a.pgm f0b.pgm [f1a.pgm f1b.pgm ...]\n");
        return 2;
    }
    const int n_pairs = (int)files.size() / 2;
    int ndev = 0;
    HIP_OK(hipGetDeviceCount(&ndev));
    if (gpus > ndev) { fprintf(stderr, "--gpus %d but %d device(s) visible\n", gpus, ndev); return 1; }

    int w = 0, h = 0;
    if (!read_pgm(files[0], nullptr, w, h)) { fprintf(stderr, "Could not open %s\n", files[0]); return 1; }
    // three rounds of pinned pair slots, filled ahead of the round loop by a reader thread
    FrameRing ring;
    ring.start(files, w, h, 3 * gpus);
    bbme_params params{};
    params.num_levels = levels;
    for (int l = 0; l < levels; ++l) { params.block_size[l] = block; params.search_size[l] = search; }

    HipBackend be;
    be.gpus = gpus; be.n_pairs = n_pairs; be.w = w; be.h = h; be.frames = &ring; be.out_dir = out_dir;
    std::vector<int> devs(gpus);
    for (int r = 0; r < gpus; ++r) devs[r] = r;
    be.comms.resize(gpus);
    NCCL_OK(ncclCommInitAll(be.comms.data(), gpus, devs.data()));
    be.ctx.assign(gpus, nullptr);
    be.stream.assign(gpus, nullptr);
    for (int r = 0; r < gpus; ++r) {
        BBME_OKAY(bbme_create(&params, w, h, r, &be.ctx[r]));
        // no speculative search: its graph has a forked branch, and a second stream (the copy stream here) waiting for an event
        // behind such a graph costs the next replay 0.7 ms -- more than the speculation gains
        if (!getenv("BBME_SPECULATE")) BBME_OKAY(bbme_set_speculation(be.ctx[r], 0));
        void *s = nullptr;
        BBME_OKAY(bbme_get_stream(be.ctx[r], &s));
        be.stream[r] = static_cast<hipStream_t>(s);
    }
    BBME_OKAY(bbme_get_geometry(be.ctx[0], &be.pw, &be.ph, &be.pad_x, &be.pad_y));
    be.words = (size_t)(be.pw / 2) * (be.ph / 2);
    HIP_OK(hipSetDevice(0));
    HIP_OK(hipStreamCreateWithFlags(&be.copy_stream, hipStreamNonBlocking));
    // one round holds `gpus` files: enough rounds in the ring for every writer to have a file, plus the round being filled
    be.nbuf = std::max(2, (writers + gpus - 1) / gpus + 1);
    be.d_recv.assign(be.nbuf, nullptr); be.host.assign(be.nbuf, nullptr);
    be.ev_gathered.assign(be.nbuf, nullptr); be.ev_downloaded.assign(be.nbuf, nullptr);
    be.downloaded_once.assign(be.nbuf, 0);
    for (int b = 0; b < be.nbuf; ++b) {
        HIP_OK(hipMalloc(&be.d_recv[b], be.words * gpus * sizeof(int32_t)));
        // pinned staging for the writer: the gathered cell grids of a round (1/16 of the dense fields); the writer expands them
        // while it writes (bbme_flo_writer_submit_cells)
        HIP_OK(hipHostMalloc(&be.host[b], be.words * gpus * sizeof(int32_t)));
        HIP_OK(hipEventCreateWithFlags(&be.ev_gathered[b], hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&be.ev_downloaded[b], hipEventDisableTiming));
    }
    BBME_OKAY(bbme_flo_writer_create_pool(writers, &be.writer));

    const auto t0 = std::chrono::steady_clock::now();
    bbme::run_sequence(be, gpus, n_pairs, 0, be.nbuf);
    for (int r = 0; r < gpus; ++r) BBME_OKAY(bbme_synchronize(be.ctx[r]));       // also refuses a field that did not converge
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%d pairs of %dx%d on %d GPU(s), %d writer(s), %d staging buffers: %.3f s (%.2f ms per pair, files included)\n", n_pairs, w, h, gpus,
           writers, be.nbuf, secs, secs / n_pairs * 1e3);
    // per-round phases on GPU 0 (device time between stamps; the download runs on the copy stream beside the next round)
    const size_t rounds = be.t_gather.size();
    for (size_t k = 0; k < rounds && k < be.t_start.size(); ++k) {
        float up = 0, est = 0, ga = 0, dl = 0, cycle = 0;
        HIP_OK(hipEventElapsedTime(&up, be.t_start[k], be.t_frames[k]));
        HIP_OK(hipEventElapsedTime(&est, be.t_frames[k], be.t_estimate[k]));
        HIP_OK(hipEventElapsedTime(&ga, be.t_estimate[k], be.t_gather[k]));
        HIP_OK(hipEventElapsedTime(&dl, be.t_gather[k], be.t_download[k]));
        if (k + 1 < be.t_start.size()) HIP_OK(hipEventElapsedTime(&cycle, be.t_start[k], be.t_start[k + 1]));
        printf("round %zu: upload+pyramid %.3f ms, estimate %.3f ms, gather %.3f ms, download (copy stream, from the gather's end) %.3f ms, "
               "next round starts after %.3f ms\n", k, up, est, ga, dl, cycle);
    }

    for (auto *v : {&be.t_start, &be.t_frames, &be.t_estimate, &be.t_gather, &be.t_download})
        for (hipEvent_t e : *v) (void)hipEventDestroy(e);
    bbme_flo_writer_destroy(be.writer);
    ring.finish();
    for (int b = 0; b < be.nbuf; ++b) { (void)hipHostFree(be.host[b]); (void)hipFree(be.d_recv[b]); }
    for (int r = 0; r < gpus; ++r) { bbme_destroy(be.ctx[r]); ncclCommDestroy(be.comms[r]); }
    return 0;
}
