#!/usr/bin/env python3
"""Workload for a kernel-trace timeline of several pairs in flight on ONE GPU (the N = 1 point of the 8-pair batch):

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 scripts/seq_workload.py --pairs 8 --steps 4
    python scripts/seq_timeline.py OUT

One context (and private stream) per pair, speculation and relaxation as bench.py's `sequence` leg sets them."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg3")
ap.add_argument("--pairs", type=int, default=8)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--speculate", type=int, default=0)
ap.add_argument("--relax", type=int, default=1)
ap.add_argument("--batch", type=int, default=1, help="pairs per batched context (bbme_create_batch)")
a = ap.parse_args()

import blockbasedmotionestimation_amd as bbme                    # noqa: E402
from bench import WORKLOADS                                      # noqa: E402

w, h, search, block, levels, _ = WORKLOADS[a.workload]
ctxs = []
import numpy as np                                               # noqa: E402
frames = [bbme.synth_pair(w, h, 1030 + k, max_motion=24)[:2] for k in range(min(a.pairs, 8))]
while len(frames) < a.pairs:       # deeper than 8: the same pairs rolled by a few pixels, as bench.py's sequence_deep leg does
    k = len(frames)
    sh = (3 * (k // 8), 5 * (k // 8))
    frames.append(tuple(np.ascontiguousarray(np.roll(f, sh, (0, 1))) for f in frames[k % 8]))
for i in range(0, a.pairs, a.batch):
    mf = bbme.MFBatch(frames[i:i + a.batch], [search] * levels, [block] * levels, levels)
    mf.set_speculation(bool(a.speculate))
    mf.set_relaxation(bool(a.relax))
    ctxs.append(mf)
for c in ctxs:
    c.estimate_async()
for c in ctxs:
    c.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    for c in ctxs:
        c.estimate_async()
for c in ctxs:
    c.synchronize()
dt = time.perf_counter() - t0
blocks0 = (ctxs[0].padded_width // block) * (ctxs[0].padded_height // block)
print("seq workload: %d pairs (%d contexts x %d) x %d steps, %.3f ms per pair, %.2f Mblocks/s" %
      (a.pairs, len(ctxs), a.batch, a.steps, dt / (a.pairs * a.steps) * 1e3, blocks0 * a.pairs * a.steps / dt / 1e6))
for c in ctxs:
    c.close()
