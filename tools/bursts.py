#!/usr/bin/env python3
"""Per-stream activity bursts from a rocprofv3 kernel trace of bench.py (ms since a BEV conv start)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp'])
    r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
marks = [r['s'] for r in rows if 'small_cin' in r['Kernel_Name'] and (', 6>' in r['Kernel_Name'] or ', 6, ' in r['Kernel_Name'])]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t0 = marks[first]
print('bev conv starts (ms):', [round((m - t0) / 1e6, 2) for m in marks[first:first + 6]])
for sid in sorted({r['Stream_Id'] for r in rows}):
    rs = [r for r in rows if r['Stream_Id'] == sid and t0 <= r['s'] < marks[first + 5]]
    bursts = []
    for r in rs:
        if bursts and r['s'] - bursts[-1][1] < 60e3:
            bursts[-1][1] = max(bursts[-1][1], r['e'])
            bursts[-1][2] += 1
        else:
            bursts.append([r['s'], r['e'], 1])
    print('stream', sid, ' '.join('[%.2f-%.2f n%d]' % ((a - t0) / 1e6, (b - t0) / 1e6, n)
                                  for a, b, n in bursts))
