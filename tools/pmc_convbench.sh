# FETCH / WRITE / MFMA-busy of the conv kernels of tools/conv_bench.py (each net alone): bash tools/pmc_convbench.sh <tag> [env ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
for c in FETCH_SIZE WRITE_SIZE; do
  env "$@" timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/${tag}_$c -o $tag -- python3 tools/conv_bench.py 3 > gpurun_out/${tag}_$c.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for c, corr in (('FETCH_SIZE', 2.0), ('WRITE_SIZE', 1.0)):
    path = glob.glob('gpurun_out/${tag}_%s/**/*counter_collection.csv' % c, recursive=True)[0]
    disp = collections.defaultdict(lambda: [None, 0.0])
    for r in csv.DictReader(open(path)):
        d = disp[r['Dispatch_Id']]; d[0] = r['Kernel_Name']; d[1] += float(r['Counter_Value'])
    for name, v in disp.values():
        k = name.split('(')[0].split('::')[-1][:32]
        tot[k][c][0] += v * 1024 * corr; tot[k][c][1] += 1
for k, d in tot.items():
    if 'conv' in k or 'wino' in k:
        print('%-34s fetch %7.1f MB  write %6.1f MB per launch (%d launches)' % (k, d['FETCH_SIZE'][0] / max(d['FETCH_SIZE'][1], 1) / 1e6, d['WRITE_SIZE'][0] / max(d['WRITE_SIZE'][1], 1) / 1e6, d['FETCH_SIZE'][1]))
PY
