#!/usr/bin/env python3
"""ms per pair against the number of pairs in flight on one GPU (one context + stream per pair), exact and Jacobi regulariser,
with and without a staggered start.  Development aid for the `sequence` leg of bench.py."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg3")
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--pairs", default="1,2,3,4,6,8,12")
a = ap.parse_args()
import blockbasedmotionestimation_amd as bbme                    # noqa: E402
from bench import WORKLOADS                                      # noqa: E402
w, h, search, block, levels, _ = WORKLOADS[a.workload]
maxp = max(int(x) for x in a.pairs.split(","))
ctxs = []
for k in range(maxp):
    f1, f2, _ = bbme.synth_pair(w, h, 1030 + k, max_motion=24)
    ctxs.append(bbme.MF(f1, f2, [search] * levels, [block] * levels, levels))
blocks0 = (ctxs[0].padded_width // block) * (ctxs[0].padded_height // block)


def run(cs, steps, stagger_s=0.0):
    for c in cs:
        c.estimate_async()
    for c in cs:
        c.synchronize()
    t0 = time.perf_counter()
    if stagger_s:
        for c in cs:                       # first pyramid of every stream offset in time; the rest queue behind it
            c.estimate_async()
            t = time.perf_counter()
            while time.perf_counter() - t < stagger_s:
                pass
        for _ in range(steps - 1):
            for c in cs:
                c.estimate_async()
    else:
        for _ in range(steps):
            for c in cs:
                c.estimate_async()
    for c in cs:
        c.synchronize()
    return (time.perf_counter() - t0) / (steps * len(cs)) * 1e3


for mode, jac, spec, relax in (("exact", False, False, False), ("exact+relax", False, False, True),
                               ("exact+spec+relax", False, True, True), ("jacobi", True, False, False)):
    for c in ctxs:
        c.set_regularizer_mode(jac)
        c.set_speculation(spec)
        c.set_relaxation(relax)
    line = []
    for p in (int(x) for x in a.pairs.split(",")):
        ms = run(ctxs[:p], a.steps)
        line.append("P=%d %.3f" % (p, ms))
    print("%-18s ms/pair: %s" % (mode, "  ".join(line)), flush=True)
# the same pairs as batched contexts (bbme_create_batch): n contexts x b pairs each
for c in ctxs:
    c.close()
ctxs = []
frames = [bbme.synth_pair(w, h, 1030 + k, max_motion=24)[:2] for k in range(16)]
for total in (8, 16):
    for b in (1, 2, 4, 8, 16):
        if b > total:
            continue
        bs = [bbme.MFBatch(frames[i * b:(i + 1) * b], [search] * levels, [block] * levels, levels) for i in range(total // b)]
        for spec, relax in ((False, False), (False, True), (True, True)):
            for c in bs:
                c.set_speculation(spec)
                c.set_relaxation(relax)
            ms = run(bs, max(3, a.steps // 2)) / b
            print("batched: %2d pairs as %2d contexts x %2d pairs, spec %d relax %d: %.3f ms/pair  (%.1f Mblocks/s)" %
                  (total, total // b, b, spec, relax, ms, blocks0 / ms / 1e3), flush=True)
        for c in bs:
            c.close()
sys.exit(0)
for stag in (100e-6, 200e-6, 400e-6):
    print("exact, P=8, stagger %3.0f us: %.3f ms/pair" % (stag * 1e6, run(ctxs[:8], a.steps, stag)), flush=True)
for c in ctxs:
    c.close()
