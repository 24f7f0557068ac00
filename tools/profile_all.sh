# Round-3 evidence set: plain bench runs + rocprofv3 passes of the same commands (tools/profile_r3.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err &&
bash tools/profile_r3.sh r3 &&
timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt --conv-dtype bf16 --head-dtype bf16 > gpurun_out/r3bf16_bench.json 2>/dev/null &&
bash tools/profile_r3.sh r3bf16 --conv-dtype bf16 --head-dtype bf16 &&
DODT_CONV_WINO=4 timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt > gpurun_out/r3w4_bench.json 2>/dev/null
tail -c 300 gpurun_out/r3_bench.json
