# PMC breakdown of the conv kernels of tools/conv_bench.py (each net alone): usage
#   bash tools/pmc_conv.sh <tag> [dtype]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; dt=${2:-f32}
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_pmc1 -o $tag -- python3 tools/conv_bench.py 3 $dt > gpurun_out/${tag}_pmc1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_pmc2 -o $tag -- python3 tools/conv_bench.py 3 $dt > gpurun_out/${tag}_pmc2.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_trace -o $tag -- python3 tools/conv_bench.py 5 $dt > gpurun_out/${tag}_trace.log 2>&1
tail -3 gpurun_out/${tag}_trace.log
