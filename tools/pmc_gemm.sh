# PMC breakdown of fc_dma_kernel alone (tools/gemm_one.py): bash tools/pmc_gemm.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_s1 -o $tag -- python3 tools/gemm_one.py > gpurun_out/${tag}_s1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_s2 -o $tag -- python3 tools/gemm_one.py > gpurun_out/${tag}_s2.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_s3 -o $tag -- python3 tools/gemm_one.py > gpurun_out/${tag}_s3.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ('s1','s2','s3'):
    for path in glob.glob('gpurun_out/${tag}_%s/**/*counter_collection.csv' % sub, recursive=True):
        disp = collections.defaultdict(dict)
        for r in csv.DictReader(open(path)):
            if 'fc_dma' not in r['Kernel_Name']: continue
            disp[r['Dispatch_Id']][r['Counter_Name']] = disp[r['Dispatch_Id']].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        n = len(disp)
        tot = collections.defaultdict(float)
        for d in disp.values():
            for k, v in d.items(): tot[k] += v
        print(sub, n, 'dispatches:', {k: round(v / max(n, 1), 1) for k, v in tot.items()})
PY
