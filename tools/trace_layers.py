#!/usr/bin/env python3
"""Per-layer table from a rocprofv3 kernel_trace.csv of tools/conv_bench.py."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
BEV = [('conv1_1', 1.95), ('conv1_2', 10.38), ('pool', 0), ('conv2_1', 5.19), ('conv2_2', 10.38),
       ('pool', 0), ('conv3_1', 5.19), ('conv3_2', 10.38), ('conv3_3', 10.38), ('pool', 0),
       ('conv4_1', 5.19), ('conv4_2', 10.38), ('conv4_3', 10.38), ('upconv3', 5.19),
       ('fusion3', 10.38), ('upconv2', 2.6), ('fusion2', 10.38), ('upconv1', 2.6),
       ('fusion1', 20.76), ('bneck', 0)]
IMG = [(n, g * (0.75 / 1.95 if n == 'conv1_1' else 432000.0 / 563200.0)) for n, g in BEV]
names = [r['Kernel_Name'] for r in rows]
bev_idx = [i for i, n in enumerate(names) if 'small_cin' in n and '6>' in n]
img_idx = [i for i, n in enumerate(names) if 'small_cin' in n and '4>' in n]
for label, layers, start in (('BEV', BEV, bev_idx[-1]), ('IMG', IMG, img_idx[-1])):
    tot = 0.0
    for k, (nm, gf) in enumerate(layers):
        r = rows[start + k]
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        tot += d
        m = re.search(r'<(.*)>', r['Kernel_Name'])
        print('%s %-8s %-26s %8.1f us %6.1f TF  grid %6d x %s  vgpr %s+%s lds %s' % (
            label, nm, m.group(1) if m else r['Kernel_Name'][23:45], d,
            2 * gf / d * 1e3 if gf else 0, int(r['Grid_Size_X']) // 256, r['Grid_Size_Y'],
            r['VGPR_Count'], r['Accum_VGPR_Count'], r['LDS_Block_Size']))
    print('%s total %.1f us' % (label, tot))
