# Round-4 profile sets (rocprofv3 with the program itself after --; counters in their own passes), one gpurun call:
#   bash tools/profile_r4.sh        -> gpurun_out/{r4,r4bf16,r4_dense}_{stats,pmc_mfma,pmc_fetch,pmc_write}, *_bench.json
# then, in the build container:  python3 tools/summarize_r4.py r4;  python3 tools/summarize_r4.py r4bf16 --dtype bf16
#   --conv-dtype bf16 --head-dtype bf16;  python3 tools/summarize_r4.py r4_dense --dense --points 300000 --proposals 4096 --tau 3 --boxes 40
set -o pipefail
python3 bench.py --steps 40 --warmup 10 > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err &&
bash tools/profile_r3.sh r4 &&
python3 bench.py --no-cpu-baseline --no-alt --steps 100 --warmup 10 --conv-dtype bf16 --head-dtype bf16 > gpurun_out/r4bf16_bench.json 2> gpurun_out/r4bf16_bench.err &&
bash tools/profile_r3.sh r4bf16 --conv-dtype bf16 --head-dtype bf16 &&
python3 bench.py --no-cpu-baseline --no-alt --steps 60 --warmup 10 --points 300000 --proposals 4096 --tau 3 --boxes 40 > gpurun_out/r4_dense_bench.json 2> gpurun_out/r4_dense_bench.err &&
bash tools/profile_r3.sh r4_dense --points 300000 --proposals 4096 --tau 3 --boxes 40
echo "profile_r4 rc $?"
