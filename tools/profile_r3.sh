# Round-3 profile set of one bench.py command: usage  bash tools/profile_r3.sh <tag> [bench.py args]
# (rocprofv3 with the program itself after --; counters in their own passes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o $tag -- python3 bench.py --no-cpu-baseline --no-alt --steps 10 "$@" > gpurun_out/${tag}_stats.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_pmc_mfma -o $tag -- python3 bench.py --no-cpu-baseline --no-alt --steps 4 --warmup 1 "$@" > gpurun_out/${tag}_pmc_mfma.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pmc_fetch -o $tag -- python3 bench.py --no-cpu-baseline --no-alt --steps 4 --warmup 1 "$@" > gpurun_out/${tag}_pmc_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pmc_write -o $tag -- python3 bench.py --no-cpu-baseline --no-alt --steps 4 --warmup 1 "$@" > gpurun_out/${tag}_pmc_write.log 2>&1
echo "profile $tag rc $?"
