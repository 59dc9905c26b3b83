#!/usr/bin/env python3
"""Where the time of one multi-GPU step (sequence.CellGather, world size 1 rehearsal) goes: host enqueue time per call and
device time per step, with parts of the step left out.  Development aid for bench.py --gpus N.

    python scripts/dist_step_probe.py [--steps 40]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--workload", default="cfg3")
a = ap.parse_args()
import torch                                                     # noqa: E402
import torch.distributed as dist                                 # noqa: E402
import blockbasedmotionestimation_amd as bbme                    # noqa: E402
from bench import WORKLOADS                                      # noqa: E402
from blockbasedmotionestimation_amd.sequence import mf_cell_gather  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
sys.stdout.flush()
saved = os.dup(1)
os.dup2(2, 1)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
w, h, search, block, levels, _ = WORKLOADS[a.workload]
f1, f2, _ = bbme.synth_pair(w, h, 1030, max_motion=24)
mf = bbme.MF(torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda(), [search] * levels, [block] * levels, levels, device=0,
             frames_on_device=True)
mf.synchronize()
ws = torch.cuda.Stream(device=0)
torch.cuda.set_stream(ws)
g = mf_cell_gather(mf, 0)


def timed(name, fn, fence):
    for _ in range(3):
        fn()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        fn()
    t1 = time.perf_counter()
    fence()
    t2 = time.perf_counter()
    os.write(saved, ("%-46s host enqueue %.3f ms/step, wall %.3f ms/step\n" % (name, (t1 - t0) / a.steps * 1e3,
                                                                              (t2 - t0) / a.steps * 1e3)).encode())


def fence():
    g.fence()
    mf.synchronize()
    torch.cuda.synchronize()


timed("estimate only (context on torch's stream)", mf.estimate_async, fence)
timed("CellGather.step (estimate, stage, gather, expand)", g.step, fence)


def step_no_gather():
    b = g.steps & 1
    g.steps += 1
    g.work_stream.wait_event(g.ev_free[b])
    g.estimate()
    with torch.cuda.stream(g.work_stream):
        g.stage[b].copy_(g.cells)
    g.ev_ready[b].record(g.work_stream)
    with torch.cuda.stream(g.side_stream):
        g.side_stream.wait_event(g.ev_ready[b])
        g.expand(g.stage[b], g.flows[0], g.side_stream.cuda_stream)
        g.ev_free[b].record(g.side_stream)


timed("the step without dist.gather", step_no_gather, fence)


def gather_only():
    with torch.cuda.stream(g.side_stream):
        dist.gather(g.stage[0], g.gather_list, dst=0)


timed("dist.gather alone", gather_only, fence)


def est_copy():
    g.estimate()
    with torch.cuda.stream(g.work_stream):
        g.stage[0].copy_(g.cells)


def est_record():
    g.estimate()
    g.ev_ready[0].record(g.work_stream)


def est_record_sidewait():
    g.estimate()
    g.ev_ready[0].record(g.work_stream)
    g.side_stream.wait_event(g.ev_ready[0])
    g.ev_free[0].record(g.side_stream)


def est_wait():
    g.work_stream.wait_event(g.ev_free[0])
    g.estimate()


def est_expand_side():
    g.estimate()
    g.expand(g.stage[0], g.flows[0], g.side_stream.cuda_stream)


timed("estimate + staging copy", est_copy, fence)
timed("estimate + event record", est_record, fence)
timed("estimate + record, side stream waits + records", est_record_sidewait, fence)
timed("wait for a side-stream event + estimate", est_wait, fence)
timed("estimate + expansion on the side stream", est_expand_side, fence)


def est_record_sidewait_only():
    g.estimate()
    g.ev_ready[0].record(g.work_stream)
    g.side_stream.wait_event(g.ev_ready[0])


hp = torch.cuda.Stream(device=0, priority=-1)


def est_record_hp_sidewait():
    g.estimate()
    g.ev_ready[0].record(g.work_stream)
    hp.wait_event(g.ev_ready[0])
    g.ev_free[0].record(hp)


def one_stream_step():
    g.estimate()
    with torch.cuda.stream(g.work_stream):
        dist.gather(g.cells, g.gather_list, dst=0)
        g.expand(g.gather_list[0], g.flows[0], g.work_stream.cuda_stream)


timed("estimate + record, side stream waits (no record)", est_record_sidewait_only, fence)
timed("estimate + record, high-priority stream waits + records", est_record_hp_sidewait, fence)
timed("one stream: estimate, gather, expand", one_stream_step, fence)
mf.set_speculation(False)
timed("no speculation: estimate only", mf.estimate_async, fence)
timed("no speculation: estimate + record, side waits + records", est_record_sidewait, fence)
dist.destroy_process_group()
