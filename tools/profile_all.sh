cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { tag=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o $tag -- python3 bench.py --no-cpu-baseline --no-alt --steps 10 "$@" > gpurun_out/${tag}_stats.log 2>&1 &&
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_pmc_mfma -o $tag -- python3 bench.py --no-cpu-baseline --no-alt --steps 4 --warmup 1 "$@" > gpurun_out/${tag}_pmc_mfma.log 2>&1 &&
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pmc_fetch -o $tag -- python3 bench.py --no-cpu-baseline --no-alt --steps 4 --warmup 1 "$@" > gpurun_out/${tag}_pmc_fetch.log 2>&1 &&
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pmc_write -o $tag -- python3 bench.py --no-cpu-baseline --no-alt --steps 4 --warmup 1 "$@" > gpurun_out/${tag}_pmc_write.log 2>&1
}
timeout -k 10 500 python bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err &&
run r2 &&
timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt --conv-dtype f32s > gpurun_out/r2f32s_bench.json 2>/dev/null &&
run r2f32s --conv-dtype f32s &&
timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt --conv-dtype bf16 --head-dtype bf16 > gpurun_out/r2bf16_bench.json 2>/dev/null &&
run r2bf16 --conv-dtype bf16 --head-dtype bf16
tail -1 gpurun_out/r2_bench.json | cut -c1-200
