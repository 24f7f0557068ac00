#!/usr/bin/env python3
"""Stream-level timeline of a rocprofv3 --kernel-trace csv of bench.py: per stream busy time,
union busy time and the biggest gaps inside the steady-state window."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp'])
    r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
# steady state: from the 3rd-last to the last first-layer BEV kernel
marks = [r['s'] for r in rows if 'small_cin' in r['Kernel_Name'] and (', 6>' in r['Kernel_Name'] or ', 6, ' in r['Kernel_Name'])]
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0     # marks to drop at the end (roofline reps)
t0, t1 = marks[-n_steps - 1 - skip], marks[-1 - skip]
win = [r for r in rows if r['s'] >= t0 and r['s'] < t1]
print('window %.3f ms, %d steps -> %.3f ms/step, %d kernels' % ((t1 - t0) / 1e6, n_steps,
                                                                  (t1 - t0) / 1e6 / n_steps, len(win)))
by_stream = defaultdict(list)
for r in win:
    by_stream[r.get('Stream_Id', r.get('Queue_Id'))].append(r)
for sid, rs in sorted(by_stream.items(), key=lambda kv: -sum(r['e'] - r['s'] for r in kv[1])):
    busy = sum(r['e'] - r['s'] for r in rs)
    print('stream %-6s %4d kernels busy %.3f ms (%.0f%%)' % (sid, len(rs), busy / 1e6,
                                                             100.0 * busy / (t1 - t0)))
# union
ev = sorted([(r['s'], 1) for r in win] + [(r['e'], -1) for r in win])
depth, last, idle, gaps = 0, t0, 0, []
for t, d in ev:
    if depth == 0 and t > last:
        idle += t - last
        gaps.append((t - last, last))
    depth += d
    last = t
print('GPU idle inside the window: %.3f ms (%.1f%%)' % (idle / 1e6, 100.0 * idle / (t1 - t0)))
# per-kernel totals
tot = defaultdict(lambda: [0, 0])
for r in win:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:70]
    tot[k][0] += r['e'] - r['s']
    tot[k][1] += 1
for k, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:22]:
    print('%9.1f us/step %5.1f calls/step  %s' % (t / 1e3 / n_steps, c / n_steps, k))

if len(sys.argv) > 4:      # detailed listing of one step
    import re
    a, b = marks[-2 - skip], marks[-1 - skip]
    for r in rows:
        if a <= r['s'] < b or a <= r['e'] < b:
            name = re.sub(r'\(anonymous namespace\)::|void |dodt::', '', r['Kernel_Name'])
            name = re.sub(r'\(.*', '', name)[:60]
            print('%8.1f %8.1f  s%s  %7.1f us  grid %6d  %s' % (
                (r['s'] - a) / 1e3, (r['e'] - a) / 1e3, r['Stream_Id'], (r['e'] - r['s']) / 1e3,
                int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), name))
