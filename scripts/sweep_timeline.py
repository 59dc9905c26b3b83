#!/usr/bin/env python3
"""Diagnostic: run cfg3 (or --workload) stage by stage and print, per sweep, the wall time
(host-timed with a sync; includes ~20 us of sync overhead) and the solver's counters."""
import argparse, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import blockbasedmotionestimation_amd as bbme

ap = argparse.ArgumentParser()
ap.add_argument("--w", type=int, default=3840); ap.add_argument("--h", type=int, default=2160)
ap.add_argument("--block", type=int, default=16); ap.add_argument("--search", type=int, default=80)
ap.add_argument("--levels", type=int, default=4)
a = ap.parse_args()
f1, f2, _ = bbme.synth_pair(a.w, a.h, 1030, max_motion=24)
mf = bbme.MF(f1, f2, [a.search] * a.levels, [a.block] * a.levels, a.levels)
for rep in range(2):
    for lvl in range(a.levels - 1, -1, -1):
        mf.synchronize(); t = time.perf_counter(); mf.stage_search(lvl); mf.synchronize()
        if rep: print("L%d search %.1f us" % (lvl, (time.perf_counter() - t) * 1e6))
        b = a.block
        while b > 1:
            for mult in (1, 2):
                mf.synchronize(); t = time.perf_counter(); mf.stage_regularize(lvl, b, mult); mf.synchronize()
                dt = (time.perf_counter() - t) * 1e6
                st = mf.sweep_stats()
                w, h, _, _ = mf.level_geometry(lvl)
                if rep: print("L%d b=%2d m=%d blocks=%8d  %7.1f us  evaluated=%8d max_rounds=%5d sum_rounds=%8d tail_passes=%d flag=%d"
                              % (lvl, b, mult, (w // b) * (h // b), dt, st[4], st[7], st[8], st[3], st[5]))
                if rep and st[13] and not st[15]:
                    print("      longest wave: %d rounds, %d of them with blocks left waiting in its queue; all waves: %d such rounds of %d"
                          % (st[13] >> 16, st[13] & 0xffff, st[14], st[8]))
                if rep and st[15]:      # a -DBBME_PHASE_PROFILE build (BBME_LIB=...): mean cycles per LANES round and phase
                    n = float(st[15])
                    print("      LANES rounds=%d  cycles/round: gather %.0f  rows+sad %.0f  smooth+argmin %.0f  store+drain %.0f  "
                          "atomics %.0f  enqueue %.0f  | sum %.0f" % ((st[15],) + tuple(st[9 + i] / n for i in range(6))
                                                                     + (sum(st[9:15]) / n,)))
                elif rep and st[9]:     # SAD memo (b >= 8): candidate look-ups of the chain rounds, how many the memo did not know, group passes
                    print("      memo: look-ups %d  misses %d (%.1f%%)  group passes %d  forwarded changes %d"
                          % (st[9], st[10], 100.0 * st[10] / st[9], st[11], st[12]))
            b >>= 1
