#!/usr/bin/env python3
"""Per-sweep kernel durations (us) of the graph-replayed pyramids in a rocprofv3 kernel trace, averaged over the
pyramids of the timed loop (the asynchronous solver's schedule differs from replay to replay):

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-iters 1
    python scripts/trace_table.py OUT

One line per level: the search, then per sweep  pass1+[relaxation]+solve."""
import csv, glob, sys
import os
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)   # the newest trace
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_expand' in r['Kernel_Name']]
# pyramids between consecutive k_expand launches; keep those with the most common kernel count (the graph replays)
spans = [(idx[i] + 1, idx[i + 1] + 1) for i in range(len(idx) - 1)]
from collections import Counter
spans = [s for s in spans if any('k_reg_solve' in rows[k]['Kernel_Name'] for k in range(*s))]   # not the Jacobi-mode pyramids
common = Counter(b - a for a, b in spans).most_common(1)[0][0]
spans = [s for s in spans if s[1] - s[0] == common][1:]          # drop the first (warm-up)
n = len(spans)
names = [rows[spans[0][0] + k]['Kernel_Name'] for k in range(common)]
dur = [sum((int(rows[a + k]['End_Timestamp']) - int(rows[a + k]['Start_Timestamp'])) / 1e3 for a, b in spans) / n for k in range(common)]
wall = sum((int(rows[b - 1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3 for a, b in spans) / n
line = []; tot = {'search': 0.0, 'pass1': 0.0, 'iter': 0.0, 'solve': 0.0}; p1 = it = 0.0
for nme, d in zip(names, dur):
    if 'search' in nme:
        if line: print(' '.join(line))
        line = ['search %6.1f |' % d]; tot['search'] += d
    elif 'pass1' in nme: tot['pass1'] += d; p1 = d; it = 0.0
    elif 'k_reg_iter' in nme: tot['iter'] += d; it += d
    elif 'solve' in nme:
        tot['solve'] += d
        line.append(('%5.1f+%5.1f' % (p1, d)) if it == 0.0 else ('%5.1f+[%.1f]+%5.1f' % (p1, it, d)))
print(' '.join(line))
print('mean of %d pyramids, total %.1f us: search %.1f pass1 %.1f relax %.1f solve %.1f' % (n, wall, tot['search'], tot['pass1'], tot['iter'], tot['solve']))
