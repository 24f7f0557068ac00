# PMC breakdown of the correlation kernel alone (tools/corr_bench.py): bash tools/pmc_corr.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_f -o $tag -- python3 tools/corr_bench.py > gpurun_out/${tag}_f.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_w -o $tag -- python3 tools/corr_bench.py > gpurun_out/${tag}_w.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_s1 -o $tag -- python3 tools/corr_bench.py > gpurun_out/${tag}_s1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_s2 -o $tag -- python3 tools/corr_bench.py > gpurun_out/${tag}_s2.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ('f','w','s1','s2'):
    for path in glob.glob('gpurun_out/${tag}_%s/**/*counter_collection.csv' % sub, recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        disp = collections.defaultdict(dict)
        for r in csv.DictReader(open(path)):
            if 'correlation' not in r['Kernel_Name']: continue
            disp[r['Dispatch_Id']][r['Counter_Name']] = disp[r['Dispatch_Id']].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        n = len(disp)
        tot = collections.defaultdict(float)
        for d in disp.values():
            for k, v in d.items(): tot[k] += v
        print(sub, n, 'dispatches:', {k: round(v / max(n, 1), 1) for k, v in tot.items()})
PY
