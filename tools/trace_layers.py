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
bev_idx = [i for i, n in enumerate(names) if re.search(r'small_cin_kernel<\d+, \d+, 6[,>]', n)]
img_idx = [i for i, n in enumerate(names) if re.search(r'small_cin_kernel<\d+, \d+, 4[,>]', n)]
def dur(r):
    return (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3


def is_tail(r):          # quarter-size tiles of a tail launch: <TW, 4, 4, 1, 32, false>
    return re.search(r'mfma_kernel<\d+, 4, 4, 1, 32, false>', r['Kernel_Name']) is not None


for label, layers, start in (('BEV', BEV, bev_idx[-1]), ('IMG', IMG, img_idx[-1])):
    convs = [(n, g) for n, g in layers if g]          # pools / bottleneck may be fused away
    tot = 0.0
    k = start
    t_first = int(rows[start]['Start_Timestamp'])
    ci = 0
    while ci < len(convs) and k < len(rows):
        r = rows[k]
        d = dur(r)
        k += 1
        if 'conv3x3' not in r['Kernel_Name']:
            tot += d
            print('%s %-8s %-26s %8.1f us' % (label, '', r['Kernel_Name'][23:49], d))
            continue
        nm, gf = convs[ci]
        ci += 1
        tail = ''
        if k < len(rows) and is_tail(rows[k]):
            gap = (int(rows[k]['Start_Timestamp']) - int(r['End_Timestamp'])) / 1e3
            tail = ' + tail %.1f us (gap %.1f, grid %d)' % (dur(rows[k]), gap,
                                                          int(rows[k]['Grid_Size_X']) // 256)
            d += dur(rows[k]) + gap
            k += 1
        tot += d
        m = re.search(r'<(.*)>', r['Kernel_Name'])
        print('%s %-8s %-26s %8.1f us %6.1f TF  grid %6d  vgpr %s%s' % (
            label, nm, m.group(1) if m else r['Kernel_Name'][23:45], d,
            2 * gf / d * 1e3 if gf else 0, int(r['Grid_Size_X']) // 256, r['VGPR_Count'], tail))
    wall = (int(rows[k - 1]['End_Timestamp']) - t_first) / 1e3
    print('%s kernels %.1f us, first start to last end %.1f us' % (label, tot, wall))
