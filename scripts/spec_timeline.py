#!/usr/bin/env python3
"""One graph replay of the pyramid from a rocprofv3 kernel trace (speculative search on): start, end, duration in us of every
search launch, with the regulariser launches between them folded into one line per group.

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --profile-iters 1 --in-flight 0 --no-host-boundary
    python scripts/spec_timeline.py OUT"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_expand' in r['Kernel_Name']]
spans = [(idx[i] + 1, idx[i + 1] + 1) for i in range(len(idx) - 1)]
spans = [s for s in spans if any('k_search_list' in rows[k]['Kernel_Name'] for k in range(*s))]    # replays with a fix-up
a, b = spans[len(spans) // 2]
t0 = int(rows[a]['Start_Timestamp'])
group = []
def flush():
    global group
    if group:
        s = min(int(r['Start_Timestamp']) for r in group); e = max(int(r['End_Timestamp']) for r in group)
        print("%9.1f %9.1f %8.1f  sweeps: %d regulariser launches" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, len(group)))
        group = []
for r in rows[a:b]:
    n = r['Kernel_Name']
    short = n.split('(')[0].replace('void ', '').replace('bbme::', '')
    if 'k_reg' in n:
        group.append(r)
        continue
    flush()
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%9.1f %9.1f %8.1f  %s  grid=%s threads" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, short, r.get('Grid_Size_X', '?')))
flush()
