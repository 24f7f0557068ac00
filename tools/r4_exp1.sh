run() { # label, env..., args
  label=$1; shift
  env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-alt --steps 300 --warmup 30 $ARGS 2>/dev/null | python3 tools/print_layers.py | sed -n 1p | sed "s/^/$label: /"
}
ARGS="--conv-dtype bf16 --head-dtype bf16"
run "bf16 base" A=1; run "bf16 items_per_cu 2" DODT_CONV_BF16_ITEMS_PER_CU=2; run "bf16 fc bk64" DODT_FC_BF16_DMA_BK=64; run "bf16 base" A=1
run "bf16 hwq 4" GPU_MAX_HW_QUEUES=4; run "bf16 hwq 6" GPU_MAX_HW_QUEUES=6; run "bf16 hwq 12" GPU_MAX_HW_QUEUES=12
ARGS=""
run "f32 base" A=1; run "f32 lookahead 0" DODT_BENCH_LOOKAHEAD=0; run "f32 base" A=1; run "f32 lookahead 0" DODT_BENCH_LOOKAHEAD=0
run "f32 hwq 4" GPU_MAX_HW_QUEUES=4; run "f32 hwq 6" GPU_MAX_HW_QUEUES=6; run "f32 hwq 12" GPU_MAX_HW_QUEUES=12
true
