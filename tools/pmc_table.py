#!/usr/bin/env python3
"""Per-kernel table of rocprofv3 --pmc counter_collection.csv files: usage
   pmc_table.py file1.csv [file2.csv ...]   (averages per kernel name over dispatches)"""
import collections
import csv
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        n = r['Kernel_Name']
        if 'conv3x3' not in n and 'fc_mfma' not in n and 'upsample' not in n and 'wino' not in n:
            continue
        m = re.search(r'(conv3x3_\w+|wino3x3_\w+|wino43_\w+|fc_mfma_kernel|fc_dma_kernel|upsample\w+)<(.*?)>', n)
        key = (m.group(1)[8:14] + '<' + m.group(2) + '>') if m else n[:40]
        key += ' g%d' % (int(r['Grid_Size']) // 256)
        acc[key][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[key][r['Counter_Name']] += 1
names = sorted({c for k in acc for c in acc[k]})
print('%-44s %s' % ('kernel', ' '.join('%10s' % c.replace('SQ_', '')[:10] for c in names)))
for k in sorted(acc):
    a = {c: acc[k][c] / max(cnt[k][c], 1) for c in names}
    line = '%-44s ' % k[:44] + ' '.join('%10.3g' % a[c] for c in names)
    if 'SQ_WAVE_CYCLES' in a and a.get('SQ_WAVE_CYCLES'):
        w = a['SQ_WAVE_CYCLES']
        line += '  | wait %.2f winst %.2f active %.2f' % (a.get('SQ_WAIT_ANY', 0) / w, a.get('SQ_WAIT_INST_ANY', 0) / w, a.get('SQ_ACTIVE_INST_ANY', 0) / w)
        if a.get('GRBM_GUI_ACTIVE'):
            line += ' mfma_util %.3f' % (a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (a['GRBM_GUI_ACTIVE'] / 8 * 1024))
    print(line)
