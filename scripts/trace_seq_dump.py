#!/usr/bin/env python3
"""Development aid: the kernel sequence of one graph-replayed pyramid from a rocprofv3 kernel trace (start, duration, name, grid)."""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_expand' in r['Kernel_Name']]
spans = [(idx[i] + 1, idx[i + 1] + 1) for i in range(len(idx) - 1)]
spans = [sp for sp in spans if any('k_reg_solve' in r['Kernel_Name'] for r in rows[sp[0]:sp[1]])]   # not the Jacobi-mode pyramids
a, b = spans[len(spans) // 2]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('bbme::', '')
    print("%9.1f %7.1f  %-40s grid=%s" % ((s - t0) / 1e3, (e - s) / 1e3, n, r.get('Grid_Size_X', r.get('Grid_Size', '?'))))
