#!/usr/bin/env python3
"""Print the first N kernels of a rocprofv3 kernel trace (csv) in start order with durations and the gap
to the previous kernel's end: python tools/trace_gaps.py <dir> [N] [skip]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
prev = None
for r in rows[skip:skip + n]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%-44s dur %7.1f us  gap %7.1f  grid %s' % (r['Kernel_Name'].replace('(anonymous namespace)::', '')[:44],
                                                      (e - s) / 1e3, (s - prev) / 1e3 if prev else 0, r.get('Grid_Size', '')))
    prev = e
