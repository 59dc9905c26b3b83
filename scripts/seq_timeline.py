#!/usr/bin/env python3
"""Chip occupancy over time from a rocprofv3 kernel trace of several pairs in flight (scripts/seq_workload.py).

    python scripts/seq_timeline.py OUT [bucket_us]

Takes the steady-state half of the trace, cuts it into buckets and prints, per bucket, how many kernels of each kind
were running (time-weighted mean) and how many workgroups they had asked for; then the totals per kind: busy time
(union over streams), summed kernel time, mean concurrency."""
import csv, glob, os, sys
from collections import defaultdict
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
bucket = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 100e3
rows = [r for r in csv.DictReader(open(f))]
def kind(n):
    for k in ('k_search_list', 'k_fixup_list', 'k_search', 'k_reg_pass1', 'k_reg_iter', 'k_reg_solve', 'k_expand'):
        if k in n: return k.replace('k_reg_', '').replace('k_', '')
    return None
ev = []
for r in rows:
    k = kind(r['Kernel_Name'])
    if k is None: continue
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    wgs = (int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0)) // max(1, int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 64)) or 64))
    ev.append((s, e, k, wgs, r.get('Queue_Id', r.get('Stream_Id', '?'))))
ev.sort()
t_lo, t_hi = ev[0][0], max(e for _, e, _, _, _ in ev)
lo = t_lo + (t_hi - t_lo) // 2            # steady state: the second half
sel = [x for x in ev if x[1] > lo]
kinds = ['search', 'search_list', 'fixup_list', 'pass1', 'iter', 'solve', 'expand']
print("trace %.1f ms, analysing the last %.1f ms; %d kernels; queues seen: %d" % ((t_hi - t_lo) / 1e6, (t_hi - lo) / 1e6, len(sel), len(set(x[4] for x in sel))))
print("%10s " % "t (us)" + " ".join("%11s" % k for k in kinds) + "   | running kernels (time-weighted mean) / their workgroups")
nb = int((t_hi - lo) / bucket) + 1
for b in range(min(nb, 40)):
    b0, b1 = lo + b * bucket, lo + (b + 1) * bucket
    run = defaultdict(float); wg = defaultdict(float)
    for s, e, k, w, q in sel:
        o = min(e, b1) - max(s, b0)
        if o > 0: run[k] += o / bucket; wg[k] += w * o / bucket
    print("%10.0f " % ((b0 - lo) / 1e3) + " ".join("%4.1f/%6.0f" % (run[k], wg[k]) for k in kinds))
# totals
span = t_hi - lo
print("\nkind        kernel-time(ms)  union-busy(ms)  share-of-span  mean-concurrency-while-busy")
for k in kinds:
    iv = sorted((max(s, lo), e) for s, e, kk, w, q in sel if kk == k)
    if not iv: continue
    tot = sum(e - s for s, e in iv)
    u = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: u += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    u += ce - cs
    print("%-12s %12.3f %14.3f %13.2f %14.2f" % (k, tot / 1e6, u / 1e6, u / span, tot / max(u, 1)))
# idle: no kernel of any kind running
iv = sorted((max(s, lo), e) for s, e, kk, w, q in sel)
u = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: u += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
u += ce - cs
print("any kernel running: %.3f ms of %.3f ms (%.1f %%)" % (u / 1e6, span / 1e6, 100.0 * u / span))
