#!/usr/bin/env python3
"""Condense rocprofv3 outputs (gpurun_out/<tag>_stats, _pmc_mfma, _pmc_fetch, _pmc_write of
one `bench.py` command) into profiles/<tag>_*.  Usage: summarize_profile.py r1"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def is_conv(name):
    return 'conv3x3' in name or 'wino3x3' in name or 'wino43' in name


tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, 'profiles')
os.makedirs(out, exist_ok=True)


def one(pattern):
    hits = glob.glob(os.path.join(root, 'gpurun_out', pattern), recursive=True)
    return hits[0] if hits else None


stats = one('%s_stats/**/*kernel_stats.csv' % tag)
shutil.copy(stats, os.path.join(out, '%s_kernel_stats.csv' % tag))
rows = list(csv.DictReader(open(stats)))
extra = ' '.join(sys.argv[2:])
lines = ['# %s: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-alt %s' % (tag, extra), '',
         '| kernel | calls | total ms | avg us | % |', '|---|---|---|---|---|']
conv_ns = conv_calls = 0
for r in rows[:24]:
    lines.append('| %s | %s | %.2f | %.1f | %s |' % (
        r['Name'][:70].replace('|', '/'), r['Calls'], float(r['TotalDurationNs']) / 1e6,
        float(r['AverageNs']) / 1e3, r['Percentage']))
for r in rows:
    if is_conv(r['Name']):
        conv_ns += float(r['TotalDurationNs'])
        conv_calls += int(r['Calls'])
lines += ['', 'conv kernels (wino43_* + wino3x3_* + conv3x3_*): %d launches, %.2f ms total, average launch %.1f us '
          '(the two nets overlap on two streams during the timed steps, which stretches '
          'each kernel)' % (conv_calls, conv_ns / 1e6, conv_ns / 1e3 / max(conv_calls, 1))]
# the roofline section of bench.py: the LAST reps forwards of each net, run alone
trace = one('%s_stats/**/*kernel_trace.csv' % tag)
if trace:
    tr = [r for r in csv.DictReader(open(trace)) if is_conv(r['Kernel_Name'])]
    tr.sort(key=lambda r: int(r['Start_Timestamp']))
    reps = 10
    # bench.py ends with `reps` forwards of each net alone, then `reps` forwards side by side
    alone = tr[-2 * reps * 32:-reps * 32]
    both = tr[-reps * 32:]
    if len(alone) == reps * 32:
        d = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in alone)
        lines += ['', "bench.py's roofline section (%d forwards of each net, run alone): %d conv "
                  'launches, average launch %.1f us, %.3f ms per step (32 launches)' % (
                      reps, len(alone), d / 1e3 / len(alone), d / 1e6 / reps)]
        t0 = min(int(r['Start_Timestamp']) for r in both)
        t1 = max(int(r['End_Timestamp']) for r in both)
        lines += ["side-by-side section (%d steps, both nets concurrently): %.3f ms per step" % (
            reps, (t1 - t0) / 1e6 / reps)]


def counters(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        e = d.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name'],
                                                 't': int(r['End_Timestamp']) - int(r['Start_Timestamp'])})
        e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    return list(d.values())


pm = one('%s_pmc_mfma/**/*counter_collection.csv' % tag)
if pm:
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for e in counters(pm):
        if not is_conv(e['name']):
            continue
        key = e['name'].split('(')[0][-60:]
        agg[key][0] += e.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)
        agg[key][1] += e.get('GRBM_GUI_ACTIVE', 0) / 8.0 * 1024      # SIMD-cycles available
        agg[key][2] += 1
    lines += ['', '## MFMA pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs))',
              '', '| kernel | dispatches | util |', '|---|---|---|']
    tb = ta = 0.0
    for k, (b, a, n) in sorted(agg.items()):
        lines.append('| %s | %d | %.3f |' % (k, n, b / max(a, 1)))
        tb += b
        ta += a
    lines.append('| all conv kernels | | %.3f |' % (tb / max(ta, 1)))
traffic = {}
for kind, ctr, corr in (('fetch', 'FETCH_SIZE', 2.0), ('write', 'WRITE_SIZE', 1.0)):
    pth = one('%s_pmc_%s/**/*counter_collection.csv' % (tag, kind))
    if not pth:
        continue
    tot = n = 0
    for e in counters(pth):
        if is_conv(e['name']):
            tot += e.get(ctr, 0) * 1024.0 * corr     # KB units; gfx950: FETCH_SIZE reads 1/2
            n += 1
    traffic[kind] = (tot, n)
if traffic:
    lines += ['', '## HBM traffic of the conv kernels (PMC, separate passes; FETCH_SIZE x2 per the '
              'gfx950 correction in MI355X_MICROARCH.md)', '']
    for k, (tot, n) in traffic.items():
        lines.append('- %s: %.1f MB over %d conv dispatches = %.2f MB per dispatch' % (
            k, tot / 1e6, n, tot / 1e6 / max(n, 1)))
if len(traffic) == 2:
    per = {k: tot / max(n, 1) for k, (tot, n) in traffic.items()}
    json.dump({'kernel': 'wino43_* + wino3x3_* + conv3x3_* (all conv launches of the timed steps)',
               'fetch_bytes_per_launch': per['fetch'], 'write_bytes_per_launch': per['write'],
               'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE x2 '
                         '(gfx950 correction), profiles/%s_summary.md' % tag},
              open(os.path.join(out, '%s_conv_traffic.json' % tag), 'w'), indent=1)
bj = os.path.join(root, 'gpurun_out', '%s_bench.json' % tag)
if os.path.exists(bj):
    txt = [l for l in open(bj) if l.startswith('{')]
    if txt:
        lines += ['', '## bench.py line of the same build', '', '```', txt[-1].strip(), '```']
open(os.path.join(out, '%s_summary.md' % tag), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines[:40]))
