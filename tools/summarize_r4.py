#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile_r3.sh (gpurun_out/<tag>_stats, _pmc_mfma, _pmc_fetch,
_pmc_write of one `bench.py` command) into profiles/<tag>_summary.md, profiles/<tag>_kernel_stats.csv,
profiles/<tag>_conv_traffic.json (per conv kernel) and, when the command ran the HBM-bound kernels'
stand-alone section, profiles/<tag>_hbm_traffic.json.  Round 4: the traffic tables are ALSO written under the
names bench.py reads -- profiles/conv_traffic_<dtype>.json, profiles/hbm_traffic.json (or, with --dense,
profiles/dense_hbm_traffic.json) -- each carrying the git head it was taken at (`head`), which bench.py prints as
`traffic_head`.  Usage: summarize_r4.py <tag> [--dtype f32|bf16|f32s] [--dense] [the bench.py args]

Counters follow MI355X_MICROARCH.md's HBM section: FETCH_SIZE and WRITE_SIZE in separate passes, in KB;
FETCH_SIZE x2 on gfx950 (128-byte requests tallied at 64)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

CONV = ('wino43_f32_kernel', 'wino3x3_f32_kernel', 'deconv3x3_dma_kernel', 'deconv3x3_f32_kernel', 'conv3x3_bf16_dma_kernel',
        'conv3x3_small_cin_kernel', 'conv3x3_mfma_kernel', 'conv3x3_bf16_stream_kernel', 'conv3x3_bf16_first2_kernel')
HBM = {'crop_kernel<4>': 'crop_kernel<4>', 'correlation_sp_kernel': 'correlation_sp_kernel',
       'vox_scatter': 'vox_scatter', 'vox_finalize': 'vox_finalize', 'nms_mask_kernel': 'nms_mask_kernel',
       'nms_scan_kernel': 'nms_scan_kernel'}


def short(name):
    for k in CONV:
        if k in name:
            return k
    for k in HBM:
        if k in name:
            return k
    return None


import subprocess

tag = sys.argv[1]
dtype, dense = 'f32', False
rest = sys.argv[2:]
if '--dtype' in rest:
    i = rest.index('--dtype')
    dtype = rest[i + 1]
    del rest[i:i + 2]
if '--dense' in rest:
    dense = True
    rest.remove('--dense')
sys.argv[2:] = rest
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
try:
    head = subprocess.check_output(['git', '-C', root, 'rev-parse', '--short=12', 'HEAD'], text=True).strip()
    if subprocess.check_output(['git', '-C', root, 'status', '--porcelain', '--', 'dodt_amd', 'bench.py'], text=True).strip():
        head += ' + uncommitted changes'
except Exception:
    head = 'unknown' 
out = os.path.join(root, 'profiles')


def one(pattern):
    hits = glob.glob(os.path.join(root, 'gpurun_out', pattern), recursive=True)
    return hits[0] if hits else None


def counters(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        e = d.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name'], 'id': int(r['Dispatch_Id'])})
        e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    return list(d.values())


stats = one('%s_stats/**/*kernel_stats.csv' % tag)
shutil.copy(stats, os.path.join(out, '%s_kernel_stats.csv' % tag))
rows = list(csv.DictReader(open(stats)))
lines = ['# %s: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-alt --steps 10 %s   (head %s)'
         % (tag, ' '.join(sys.argv[2:]), head), '', '| kernel | calls | total ms | avg us | % |', '|---|---|---|---|---|']
for r in rows[:26]:
    lines.append('| %s | %s | %.2f | %.1f | %s |' % (r['Name'][:70].replace('|', '/'), r['Calls'],
                                                    float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3,
                                                    r['Percentage']))
# stand-alone sections of bench.py, from the kernel trace: the LAST launches of each kernel are the
# roofline section's (each net alone, `reps` timed forwards with an event pair around every layer)
trace = one('%s_stats/**/*kernel_trace.csv' % tag)
alone = {}
if trace:
    tr = list(csv.DictReader(open(trace)))
    tr.sort(key=lambda r: int(r['Start_Timestamp']))
    bj = os.path.join(root, 'gpurun_out', '%s_stats.log' % tag)
    line = [l for l in open(bj) if l.startswith('{')]
    bench = json.loads(line[-1]) if line else None
    if bench:
        lines += ['', '## stand-alone durations: rocprofv3 kernel trace against bench.py\'s HIP events', '',
                  '| kernel | launches per step | trace avg us (stand-alone section) | bench.py avg_launch_us |', '|---|---|---|---|']
        reps = 10
        # order at the end of the run: reps plain forwards per net, reps timed forwards per net, reps side by side
        convs = [r for r in tr if short(r['Kernel_Name']) in CONV]
        per_step = sum(k['launches_per_step'] for k in bench['roofline']['kernels'])
        timed = convs[-2 * reps * per_step:-reps * per_step]
        for k in bench['roofline']['kernels']:
            sel = [r for r in timed if short(r['Kernel_Name']) == k['kernel']]
            if sel:
                avg = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in sel) / 1e3 / len(sel)
                alone[k['kernel']] = avg
                lines.append('| %s | %d | %.2f (%d launches) | %.2f |' % (k['kernel'], k['launches_per_step'], avg, len(sel),
                                                                      k['avg_launch_us']))
        for h in bench['roofline'].get('hbm', []):
            names = [n for n in HBM if n in h['kernel']] or ([n for n in HBM if n.startswith('nms_')] if h['kernel'] == 'nms_*' else [])
            tot = 0.0
            for n in names:
                sel = [r for r in tr if HBM[n] in r['Kernel_Name']][-20:]
                tot += sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in sel) / 1e3 / max(len(sel), 1)
            lines.append('| %s | - | %.2f (kernels only, last 20 launches) | %.2f (incl. memset / launch gaps) |'
                         % (h['kernel'], tot, h['us']))
pm = one('%s_pmc_mfma/**/*counter_collection.csv' % tag)
if pm:
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for e in counters(pm):
        k = short(e['name'])
        if k not in CONV:
            continue
        agg[k][0] += e.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)
        agg[k][1] += e.get('GRBM_GUI_ACTIVE', 0) / 8.0 * 1024      # SIMD-cycles available
        agg[k][2] += 1
    lines += ['', '## MFMA pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs)), all dispatches of the run',
              '', '| kernel | dispatches | util |', '|---|---|---|']
    tb = ta = 0.0
    for k, (b, a, n) in sorted(agg.items()):
        lines.append('| %s | %d | %.3f |' % (k, n, b / max(a, 1)))
        tb += b
        ta += a
    lines.append('| all conv kernels | | %.3f |' % (tb / max(ta, 1)))
per = collections.defaultdict(dict)
for kind, ctr, corr in (('fetch', 'FETCH_SIZE', 2.0), ('write', 'WRITE_SIZE', 1.0)):
    pth = one('%s_pmc_%s/**/*counter_collection.csv' % (tag, kind))
    if not pth:
        continue
    cs = counters(pth)
    for i, e in enumerate(cs):
        k = short(e['name'])
        if k is None:
            continue
        v = e.get(ctr, 0) * 1024.0 * corr          # KB units; gfx950: FETCH_SIZE reads 1/2
        if k == 'vox_scatter':                      # + the memset in front of it (hipMemsetAsync's fill kernel)
            for j in (i - 1, i - 2):
                if j >= 0 and 'fillBuffer' in cs[j]['name']:
                    v += cs[j].get(ctr, 0) * 1024.0 * corr
                    break
        d = per[k].setdefault(kind, [0.0, 0])
        d[0] += v
        d[1] += 1
if per:
    lines += ['', '## HBM-side traffic per launch (PMC, separate passes; FETCH_SIZE x2 per the gfx950 correction in '
              'MI355X_MICROARCH.md; counters sit at the L2\'s fabric side and include Infinity-Cache hits)', '',
              '| kernel | dispatches | fetch MB / launch | write MB / launch |', '|---|---|---|---|']
    conv_json, hbm_json = {}, {}
    for k, d in sorted(per.items()):
        f = d.get('fetch', [0, 1])
        w = d.get('write', [0, 1])
        lines.append('| %s | %d | %.2f | %.2f |' % (k, f[1], f[0] / max(f[1], 1) / 1e6, w[0] / max(w[1], 1) / 1e6))
        ent = {'fetch_bytes_per_launch': f[0] / max(f[1], 1), 'write_bytes_per_launch': w[0] / max(w[1], 1),
               'dispatches': f[1]}
        if k in CONV:
            conv_json[k] = ent
        else:
            hbm_json[k] = {'fetch_bytes': ent['fetch_bytes_per_launch'], 'write_bytes': ent['write_bytes_per_launch'],
                           'dispatches': f[1]}
    src = ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `python3 bench.py --no-cpu-baseline --no-alt '
           '--steps 4 --warmup 1 %s`, FETCH_SIZE x2 (gfx950 correction), profiles/%s_summary.md' % (' '.join(sys.argv[2:]), tag))
    if conv_json:
        for name in ('%s_conv_traffic.json' % tag,) + (() if dense else ('conv_traffic_%s.json' % dtype,)):
            json.dump({'kernels': conv_json, 'source': src, 'head': head}, open(os.path.join(out, name), 'w'), indent=1)
    if 'vox_scatter' in hbm_json and 'vox_finalize' in hbm_json:
        hbm_json['hipMemsetAsync + vox_scatter + vox_finalize'] = {
            'fetch_bytes': hbm_json['vox_scatter']['fetch_bytes'] + hbm_json['vox_finalize']['fetch_bytes'],
            'write_bytes': hbm_json['vox_scatter']['write_bytes'] + hbm_json['vox_finalize']['write_bytes'],
            'dispatches': hbm_json['vox_scatter']['dispatches']}
    if 'nms_mask_kernel' in hbm_json and 'nms_scan_kernel' in hbm_json:
        hbm_json['nms_*'] = {
            'fetch_bytes': hbm_json['nms_mask_kernel']['fetch_bytes'] + hbm_json['nms_scan_kernel']['fetch_bytes'],
            'write_bytes': hbm_json['nms_mask_kernel']['write_bytes'] + hbm_json['nms_scan_kernel']['write_bytes'],
            'dispatches': hbm_json['nms_mask_kernel']['dispatches'],
            'note': 'mask + scan kernels only, averaged over ALL their dispatches of the run (both NMS stages)'}
    if hbm_json:
        for name in ('%s_hbm_traffic.json' % tag, 'dense_hbm_traffic.json' if dense else ('hbm_traffic.json' if dtype == 'f32' else None)):
            if name:
                json.dump({'kernels': hbm_json, 'source': src, 'head': head}, open(os.path.join(out, name), 'w'), indent=1)
# the HBM-bound kernels on one page: bench.py's stand-alone timing + the counters of the same command
if trace and bench and bench['roofline'].get('hbm') and per:
    hl = ['# %s: HBM-bound kernels (north_star: "rocprof HBM GB/s for the voxeliser / ROI-crop")' % tag, '',
          'Stand-alone section of the plain `bench.py` run (each kernel 20 x alone on the main stream, HIP events), the kernel',
          'trace of the profiled run, and FETCH_SIZE x 2 / WRITE_SIZE of the separate `--pmc` passes (per launch).',
          'Algorithmic bytes: SURVEY.md 8(d).  Peak 8000 GB/s.', '',
          '| stage | kernel(s) | algorithmic MB | events us | trace us (kernels only) | GB/s (events) | of peak | fetch MB | write MB |',
          '|---|---|---|---|---|---|---|---|---|']
    plain = os.path.join(root, 'gpurun_out', '%s_bench.json' % tag)      # the un-profiled run's timing
    hbm_rows = bench['roofline']['hbm']
    if os.path.exists(plain):
        pl = [l for l in open(plain) if l.startswith('{')]
        if pl and json.loads(pl[-1])['roofline'].get('hbm'):
            hbm_rows = json.loads(pl[-1])['roofline']['hbm']
    for h in hbm_rows:
        names = [n for n in HBM if n in h['kernel']] or ([n for n in HBM if n.startswith('nms_')] if h['kernel'] == 'nms_*' else [])
        tot = f_mb = w_mb = 0.0
        for n in names:
            sel = [r for r in tr if HBM[n] in r['Kernel_Name']][-20:]
            tot += sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in sel) / 1e3 / max(len(sel), 1)
            f_mb += per[n].get('fetch', [0, 1])[0] / max(per[n].get('fetch', [0, 1])[1], 1) / 1e6
            w_mb += per[n].get('write', [0, 1])[0] / max(per[n].get('write', [0, 1])[1], 1) / 1e6
        hl.append('| %s | %s | %.2f | %.2f | %.2f | %.0f | %.3f | %.2f | %.2f |' % (
            h['stage'], h['kernel'], h['algorithmic_bytes'] / 1e6, h['us'], tot, h['gbps'], h['frac_of_hbm_peak'],
            f_mb, w_mb))
    hl += ['', 'Notes: the voxeliser is latency-bound (one wave round of workgroups; its memset is a separate fill kernel of '
           'the runtime: in the events column, but its 13.4 MB of writes are not a dispatch the counter pass lists); the '
           'ROI crop reads mostly L2-resident taps (its counter traffic is below the algorithmic bound of SURVEY 8d, '
           'which prices every tap); the correlation kernel (round 4: whole pixels in one pass, super-block tile curve) '
           'moves ~1.06 x its algorithmic bytes over the fabric (round 3: 1.82 x); NMS is latency-bound (single-workgroup '
           'scan), its bytes are informative only.  Taken at head %s.' % head]
    open(os.path.join(out, '%s_hbm_summary.md' % tag), 'w').write('\n'.join(hl) + '\n')
bj = os.path.join(root, 'gpurun_out', '%s_bench.json' % tag)
if os.path.exists(bj):
    txt = [l for l in open(bj) if l.startswith('{')]
    if txt:
        lines += ['', '## bench.py line of the same build (plain run, no profiler)', '', '```', txt[-1].strip(), '```']
open(os.path.join(out, '%s_summary.md' % tag), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines[:70]))
