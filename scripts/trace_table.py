#!/usr/bin/env python3
"""Per-sweep kernel durations (us) of one graph-replayed pyramid from a rocprofv3 kernel trace:

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-iters 1
    python scripts/trace_table.py OUT
"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_expand' in r['Kernel_Name']]
a, b = idx[3] + 1, idx[4] + 1
lvl = None; line = []; tot = {'solve': 0.0, 'pass1': 0.0, 'search': 0.0, 'iter': 0.0}; it = 0.0
for r in rows[a:b]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    n = r['Kernel_Name']
    if 'search' in n:
        if line: print(' '.join(line))
        line = ['search %6.1f |' % d]; tot['search'] += d
    elif 'k_reg_tile' in n: tot['pass1'] += d; p1 = d; it = 0.0
    elif 'k_reg_iter' in n: tot['iter'] += d; it += d
    elif 'solve' in n:
        tot['solve'] += d
        line.append(('%5.1f+%5.1f' % (p1, d)) if it == 0.0 else ('%5.1f+[%.1f]+%5.1f' % (p1, it, d)))
print(' '.join(line))
print('total %.1f us: search %.1f tile %.1f relax %.1f solve %.1f' % ((int(rows[b-1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3, tot['search'], tot['pass1'], tot['iter'], tot['solve']))
