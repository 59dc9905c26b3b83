// seq_schedule.hpp -- the round pipeline of a multi-GPU sequence (pair p on GPU p mod N, one gather per round),
// written once against an abstract backend: bbme_seq (csrc/bbme_seq_main.cpp) runs it over HIP streams, RCCL and the
// asynchronous .flo writer; tests/cpp/seq_schedule_test.cpp runs it over a mock that executes every queue in a random
// interleaving and checks that every file holds its own pair's result.  The same double-buffer rule is what
// sequence.CellGather (torch.distributed) follows: step k uses buffer k mod 2 and may overwrite it only behind the
// consumer of step k - 2.  Here the ring has `nbuf` >= 2 buffers: a round's files stay in its staging buffer until they
// are written, so with W writers a ring of 2 would keep at most 2 rounds' files in flight (r04: on one GPU, one file per
// round, six writers wrote no faster than two); bbme_seq sizes the ring to the writers.
//
// Nothing in the loop makes the DEVICE wait for the host: uploads, estimates, the gather and the download of round k are
// only enqueued; the host then collects round k - 1 (whose download has had a whole round to finish) and hands its grids
// to the writer.  The two host waits -- "writer done with round k - nbuf" before the download of round k may overwrite that
// staging buffer, "download of round k - 1 complete" before its files are submitted -- are on work at least one round old.
// The wait for the writer is for THAT round's files only (r04): the writer is a pool, files of later rounds may still be
// in flight, and finish in any order.  Frames live in a ring of pinned slots that a reader fills ahead of the loop: a
// round's slots are handed back once its download has completed (its uploads lie before that on the same stream).
#pragma once

namespace bbme {

// Backend concept (all enqueue calls return at once; r = rank / GPU, b = buffer 0 .. nbuf - 1, k = round):
//   void upload(int r, int pair)        frames of `pair` -> GPU r (pinned source), border + pyramid, on rank r's stream
//   void estimate(int r)                bbme_estimate on rank r's stream
//   void root_wait_downloaded(int b)    rank 0's stream waits (device side) for the last download that read receive buffer b
//   void gather(int b)                  every rank's cell grid -> receive buffer b on rank 0, on the ranks' streams
//   void record_gathered(int b)         event on rank 0's stream behind the gather
//   void host_wait_writer(int k)        HOST wait: the writer has finished every file of round k -- and only those (k < 0: nothing)
//   void release_frames(int k)          host: the frame slots of round k may be refilled (its uploads have completed)
//   void download(int b)                copy stream: waits for record_gathered(b), copies receive buffer b -> host buffer b
//   void record_downloaded(int b)       event on the copy stream behind the download
//   void host_wait_downloaded(int b)    HOST wait for record_downloaded(b)
//   void submit_files(int k, int b)     host: one writer job per pair of round k, reading host buffer b
// `faults` (tests only): bit 0 leaves out the wait for the writer, bit 1 the wait for the download, bit 2 hands a round's
// frame slots back before its uploads have run -- the mock must then catch a file with another pair's data (or an upload
// from a slot that was given away), which is what shows that the test can see a missing wait at all.
template <class Backend>
void run_sequence(Backend &be, int gpus, int n_pairs, unsigned faults = 0, int nbuf = 2)
{
    const int rounds = (n_pairs + gpus - 1) / gpus;
    for (int k = 0; k < rounds; ++k) {
        const int b = k % nbuf, bp = (k + nbuf - 1) % nbuf;           // this round's buffer, the round before's
        for (int r = 0; r < gpus; ++r) {
            const int p = k * gpus + r;
            if (p >= n_pairs) continue;                       // this rank idles in the last round but still joins the gather
            be.upload(r, p);
            be.estimate(r);
        }
        if (k >= nbuf) be.root_wait_downloaded(b);                    // the gather below overwrites receive buffer b
        be.gather(b);
        be.record_gathered(b);
        if (!(faults & 1u)) be.host_wait_writer(k - nbuf);            // the download below overwrites host buffer b
        be.download(b);
        be.record_downloaded(b);
        if (k >= 1) {                                                 // collect the round before: it has had a round to finish
            if (!(faults & 2u)) be.host_wait_downloaded(bp);
            if (!(faults & 4u)) be.release_frames(k - 1);             // (fault 4: handed back a round early, below)
            be.submit_files(k - 1, bp);
        }
        if (faults & 4u) be.release_frames(k);
    }
    if (rounds >= 1) {
        const int b = (rounds - 1) % nbuf;
        be.host_wait_downloaded(b);
        be.release_frames(rounds - 1);
        be.submit_files(rounds - 1, b);
    }
    // every round's files: rounds up to rounds - nbuf - 1 were waited for inside the loop
    for (int k = rounds > nbuf ? rounds - nbuf : 0; k < rounds; ++k) be.host_wait_writer(k);
}

}  // namespace bbme
