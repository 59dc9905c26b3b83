#!/usr/bin/env python3
"""Kernel time per pair, by kernel class, of a sequence run under `rocprofv3 --kernel-trace --stats`:

    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 scripts/seq_workload.py --pairs 24 --batch 6 --steps 8
    python3 scripts/seq_decompose.py OUT PAIRS_TIMES_STEPS_PLUS_WARMUP

Kernel DURATIONS are what the trace measures reliably (tracing serialises launches of different streams, so wall time under
the tracer says nothing); their sum per class divided by the pairs processed is the chip time a pair costs in that class when
every launch carries a whole batch.  A batched launch works on `batch` pairs at once, so "per pair" = total / (launches x batch)
x launches = total / pairs."""
import csv
import glob
import os
import sys

out, pairs = sys.argv[1], float(sys.argv[2])
path = sorted(glob.glob(os.path.join(out, "**", "*kernel_stats.csv"), recursive=True))[-1]
classes = [("search", ("k_search_fast", "k_search_generic", "k_search_list", "k_fixup_list")),
           ("pass 1", ("k_reg_pass1",)), ("relaxation", ("k_reg_iter",)), ("solver", ("k_reg_solve",)),
           ("expand", ("k_expand",)), ("padding + pyramid", ("k_pad_zero", "k_pyr_down"))]
tot = {name: [0.0, 0] for name, _ in classes}
other = [0.0, 0]
rows = list(csv.DictReader(open(path)))
for r in rows:
    ns, calls = float(r["TotalDurationNs"]), int(r["Calls"])
    for name, keys in classes:
        if any(k in r["Name"] for k in keys):
            tot[name][0] += ns; tot[name][1] += calls
            break
    else:
        other[0] += ns; other[1] += calls
print("# %s, %d pairs" % (os.path.relpath(path, out), pairs))
print("%-20s %10s %12s %14s" % ("kernel class", "launches", "total ms", "us per pair"))
s = 0.0
for name, _ in classes:
    ns, calls = tot[name]
    print("%-20s %10d %12.3f %14.1f" % (name, calls, ns / 1e6, ns / 1e3 / pairs))
    if name != "padding + pyramid":
        s += ns
print("%-20s %10s %12.3f %14.1f   (search + pass 1 + relaxation + solver + expand)" % ("estimate", "", s / 1e6, s / 1e3 / pairs))
print("%-20s %10d %12.3f" % ("other kernels", other[1], other[0] / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("    %-70s calls %6s  avg %9.1f us  total %9.3f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
