// CPU test of the ordering logic of bbme_seq's round pipeline (csrc/seq_schedule.hpp): a mock backend in which every HIP
// stream, the copy stream, every worker of the writer pool and the frame reader are queues executed in a RANDOM interleaving
// that honours only what the real machinery guarantees -- order within a queue, event waits, the root's gather completing
// after every rank's contribution, host waits.  The writer is a pool: a file goes to any worker, files finish in any order,
// and host_wait_writer(k) waits for round k's files ONLY.  Frames sit in a ring of three rounds of slots that the reader
// refills as soon as a slot is handed back; an upload copies whatever its slot holds when it RUNS.  Every file must end up
// holding its own pair's result.  With a wait left out on purpose (faults) some seed must produce a wrong file: that is what
// shows the test can see a missing wait.
//   g++ -std=c++17 -O1 -I blockbasedmotionestimation_amd/csrc tests/cpp/seq_schedule_test.cpp -o seq_schedule_test && ./seq_schedule_test
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <functional>
#include <random>
#include <vector>

#include "seq_schedule.hpp"

struct Mock {
    struct Op { std::function<bool()> ready; std::function<void()> run; };
    int gpus, n_pairs;
    int writer_lag = 0;                            // of 16 steps that could go to a writer worker, this many go elsewhere (a slow disk)
    std::mt19937 rng;
    static constexpr int kWriters = 3;
    std::vector<std::deque<Op>> q;                 // [0, gpus): rank streams; gpus: copy stream; gpus + 1 .. + kWriters: writer pool; last: frame reader
    std::vector<int> slot;                         // the frame ring: pair held by each slot, -1 = free
    std::vector<int> frames, cells;                // per rank
    static constexpr int kMaxBuf = 5;
    std::vector<int> recv[kMaxBuf], host[kMaxBuf]; // per buffer, per rank
    std::vector<int> files;                        // per pair
    std::vector<char> ticket_done;                 // events: one ticket per record call
    int gathered_ticket[kMaxBuf] = {-1, -1, -1, -1, -1}, downloaded_ticket[kMaxBuf] = {-1, -1, -1, -1, -1};
    std::vector<std::vector<int>> inflight;        // per gather: what every rank contributed
    std::vector<int> arrived;
    std::vector<int> writer_jobs_left;             // per round
    static int result_of(int pair) { return pair * 7 + 3; }

    Mock(int g, int n, unsigned seed) : gpus(g), n_pairs(n), rng(seed), q(g + 2 + kWriters), slot(3 * g, -1), frames(g, -1), cells(g, -1), files(n, -1)
    {
        for (auto &v : recv) v.assign(g, -1);
        for (auto &v : host) v.assign(g, -1);
        writer_jobs_left.assign((n + g - 1) / g + 1, 0);
        // the reader: pair after pair into its slot, as soon as that slot is free
        for (int p = 0; p < n; ++p) {
            const int s = p % (int)slot.size();
            q.back().push_back({[this, s] { return slot[s] < 0; }, [this, s, p] { slot[s] = p; }});
        }
    }
    bool step()                                     // run the head of a random runnable queue
    {
        std::vector<int> runnable;
        for (int i = 0; i < (int)q.size(); ++i)
            if (!q[i].empty() && q[i].front().ready()) runnable.push_back(i);
        if (runnable.empty()) return false;
        int i = runnable[rng() % runnable.size()];
        if (writer_lag && i > gpus && i <= gpus + kWriters && (int)(rng() % 16) < writer_lag) {
            std::vector<int> others;
            for (int j : runnable) if (j <= gpus || j > gpus + kWriters) others.push_back(j);
            if (!others.empty()) i = others[rng() % others.size()];
        }
        Op op = q[i].front();
        q[i].pop_front();
        op.run();
        return true;
    }
    void pump() { for (int n = (int)(rng() % 4); n > 0; --n) step(); }     // the device makes some progress between host calls
    void push(int stream, std::function<bool()> ready, std::function<void()> run) { q[stream].push_back({ready, run}); pump(); }
    static bool always() { return true; }
    int new_ticket() { ticket_done.push_back(0); return (int)ticket_done.size() - 1; }

    void upload(int r, int pair)
    {
        const int s = pair % (int)slot.size();
        while (slot[s] != pair) {                                           // the host waits for the reader, which runs on its own:
            auto &rq = q.back();                                            // the device need not make progress meanwhile
            if (!rq.empty() && rq.front().ready()) { Op op = rq.front(); rq.pop_front(); op.run(); }
            else if (!step()) { fprintf(stderr, "deadlock waiting for the frame reader\n"); exit(3); }
        }
        push(r, always, [this, r, s] { frames[r] = slot[s]; });              // copies what the slot holds when the copy RUNS
    }
    void release_frames(int k)
    {
        for (int r = 0; r < gpus; ++r) {
            const int p = k * gpus + r;
            if (p < n_pairs && slot[p % (int)slot.size()] == p) slot[p % (int)slot.size()] = -1;
        }
        pump();
    }
    void estimate(int r) { push(r, always, [this, r] { cells[r] = result_of(frames[r]); }); }
    void root_wait_downloaded(int b)
    {
        const int t = downloaded_ticket[b];
        if (t >= 0) push(0, [this, t] { return ticket_done[t] != 0; }, [] {});
    }
    void gather(int b)
    {
        const int id = (int)inflight.size();
        inflight.emplace_back(gpus, -1);
        arrived.push_back(0);
        for (int r = 1; r < gpus; ++r) push(r, always, [this, id, r] { inflight[id][r] = cells[r]; ++arrived[id]; });
        push(0, [this, id] { return arrived[id] == gpus - 1; },
             [this, id, b] { inflight[id][0] = cells[0]; recv[b] = inflight[id]; });
    }
    void record_gathered(int b) { const int t = new_ticket(); gathered_ticket[b] = t; push(0, always, [this, t] { ticket_done[t] = 1; }); }
    void host_wait_writer(int k)
    {
        if (k < 0) return;
        while (writer_jobs_left[k] > 0)                                     // round k's files, nothing else
            if (!step()) { fprintf(stderr, "deadlock waiting for the writer\n"); exit(3); }
    }
    void download(int b)
    {
        const int t = gathered_ticket[b];
        push(gpus, [this, t] { return ticket_done[t] != 0; }, [this, b] { host[b] = recv[b]; });
    }
    void record_downloaded(int b) { const int t = new_ticket(); downloaded_ticket[b] = t; push(gpus, always, [this, t] { ticket_done[t] = 1; }); }
    void host_wait_downloaded(int b)
    {
        const int t = downloaded_ticket[b];
        while (!ticket_done[t])
            if (!step()) { fprintf(stderr, "deadlock waiting for a download\n"); exit(3); }
    }
    void submit_files(int k, int b)
    {
        for (int r = 0; r < gpus; ++r) {
            const int p = k * gpus + r;
            if (p >= n_pairs) continue;
            ++writer_jobs_left[k];
            push(gpus + 1 + (int)(rng() % kWriters), always,
                 [this, k, b, r, p] { files[p] = host[b][r]; --writer_jobs_left[k]; });                     // reads the buffer when it RUNS
        }
    }
    void drain() { while (step()) {} }
    int wrong() const
    {
        int n = 0;
        for (int p = 0; p < n_pairs; ++p) n += files[p] != result_of(p);
        return n;
    }
};

int main()
{
    int failures = 0;
    // the pipeline as shipped: every file right, whatever the interleaving
    for (int nbuf : {2, 3, 5})
        for (int gpus : {1, 2, 3, 8})
            for (int n_pairs : {1, 2, 5, 8, 17, 40})
                for (unsigned seed = 0; seed < 100; ++seed) {
                    Mock m(gpus, n_pairs, seed * 7919u + gpus * 31u + n_pairs + nbuf * 1000003u);
                    m.writer_lag = seed % 3 == 0 ? 15 : 0;                  // every third run with writers that fall far behind
                    bbme::run_sequence(m, gpus, n_pairs, 0, nbuf);
                    m.drain();
                    if (m.wrong()) { printf("FAIL: %d buffers, %d GPUs, %d pairs, seed %u: %d wrong files\n", nbuf, gpus, n_pairs, seed, m.wrong()); ++failures; }
                }
    // a wait left out must be visible for some interleaving, whatever the depth of the ring
    for (int nbuf : {2, 4})
        for (unsigned fault : {1u, 2u, 4u}) {
            int caught = 0;
            for (unsigned seed = 0; seed < 600 && !caught; ++seed) {
                Mock m(2, 40, seed);
                m.writer_lag = fault == 1u ? 15 : 0;                        // a missing writer wait only shows when the writers are behind
                bbme::run_sequence(m, 2, 40, fault, nbuf);
                m.drain();
                caught += m.wrong() != 0;
            }
            if (!caught) { printf("FAIL: the mock never noticed fault %u with %d buffers\n", fault, nbuf); ++failures; }
        }
    if (failures) return 1;
    printf("seq_schedule ok\n");
    return 0;
}
