#!/usr/bin/env python3
"""Per-launch durations of one extractor forward from a rocprofv3 kernel trace of
tools/conv_bench.py (alone on the chip): `python tools/layer_times.py <trace.csv>`."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
convs = [r for r in rows if 'dodt::' in r['Kernel_Name']]
# the last forward of each net: the trace ends with reps x img forwards; print the last 60 launches
for r in convs[-int(sys.argv[2]) if len(sys.argv) > 2 else -60:]:
    print('%-70s %8.1f us  grid %s' % (r['Kernel_Name'][:70], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                                      r.get('Grid_Size_X', r.get('Grid_Size'))))
