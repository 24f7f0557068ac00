#!/usr/bin/env python3
"""The dependent launch chain of a frame's side stream in one steady-state step, from a rocprofv3 kernel trace
(csv) of bench.py: every kernel of that stream with its start, duration and the idle time since the previous
kernel of the same stream ended.  usage: tail_chain.py <dir or csv> [stream rank by busy time, default 2] [step from the end, default 6]"""
import csv
import glob
import sys
from collections import defaultdict

src = sys.argv[1]
f = src if src.endswith('.csv') else glob.glob(src + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
key = 'Stream_Id' if 'Stream_Id' in rows[0] else 'Queue_Id'
by = defaultdict(list)
for r in rows:
    by[r[key]].append(r)
order = sorted(by, key=lambda k: -sum(r['e'] - r['s'] for r in by[k]))
print('streams by busy time:', [(k, len(by[k]), round(sum(r['e'] - r['s'] for r in by[k]) / 1e6, 2)) for k in order])
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 2
back = int(sys.argv[3]) if len(sys.argv) > 3 else 6
rs = by[order[rank]]
# a step of a side stream starts with the voxeliser's scatter kernel
starts = [i for i, r in enumerate(rs) if 'vox_scatter' in r['Kernel_Name']]
i0, i1 = starts[-back - 1], starts[-back]
t0 = rs[i0]['s']
prev = None
tot_busy = 0
for r in rs[i0:i1]:
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('dodt::', '')[:52]
    print('%8.1f us  dur %7.1f  idle before %7.1f  %s  grid %s' % ((r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3,
                                                               (r['s'] - prev) / 1e3 if prev else 0.0, name, r.get('Grid_Size', '')))
    prev = r['e']
    tot_busy += r['e'] - r['s']
print('step of this stream: %.1f us from first start to next step\'s first start, %d kernels, busy %.1f us' % (
    (rs[i1]['s'] - t0) / 1e3, i1 - i0, tot_busy / 1e3))
