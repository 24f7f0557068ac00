# Wave-stall counters of one fully connected layer alone (tools/gemm_one.py): where do the waves of
# fc_dma_kernel (fp32, M = 1024, N = K = 2048) spend the quarter of the time the matrix pipe is idle?
#   bash tools/pmc_gemm_stalls.sh <tag> [dtype f32|bf16]  -> gpurun_out/<tag>_stalls.txt
# One counter group per pass (a pass whose counter names this rocprofv3 does not know is skipped, the others still run).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
dt=${2:-f32}
out=gpurun_out/${tag}_stalls.txt
: > $out
i=0
while read -r group; do
  i=$((i + 1))
  DODT_GEMM_ONE_DTYPE=$dt timeout -k 10 200 rocprofv3 --pmc $group --output-format csv -d gpurun_out/${tag}_g$i -o $tag -- python3 tools/gemm_one.py > gpurun_out/${tag}_g$i.log 2>&1 || echo "pass $i ($group) failed" >> $out
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE
SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS GRBM_GUI_ACTIVE
SQ_WAIT_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_FLAT SQ_WAVE_DEP_WAIT GRBM_GUI_ACTIVE
TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE
GROUPS
python3 - >> $out <<PY
import csv, glob, collections
for i in range(1, 9):
    for path in glob.glob('gpurun_out/${tag}_g%d/**/*counter_collection.csv' % i, recursive=True):
        disp = collections.defaultdict(dict)
        name = ''
        for r in csv.DictReader(open(path)):
            if 'fc_' not in r['Kernel_Name']: continue
            name = r['Kernel_Name'][:60]
            disp[r['Dispatch_Id']][r['Counter_Name']] = disp[r['Dispatch_Id']].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        n = len(disp)
        tot = collections.defaultdict(float)
        for d in disp.values():
            for k, v in d.items(): tot[k] += v
        print('pass', i, name, n, 'dispatches, per dispatch:', {k: round(v / max(n, 1), 1) for k, v in tot.items()})
PY
cat $out
