# Tail stream sets (dodt_amd/pipeline.py: tail_sets): one set = a frame's prep and tail of consecutive steps in a row
# on one stream; two sets = the steps alternate between two sets of side streams.  -> gpurun_out/r4_tailsets.txt
out=gpurun_out/r4_tailsets.txt
: > $out
for mode in "f32 f32" "bf16 f32" "bf16 bf16"; do
  set -- $mode
  for sets in 1 2; do
    DODT_PIPE_TAIL_SETS=$sets python3 bench.py --no-cpu-baseline --no-alt --steps 300 --warmup 20 --conv-dtype $1 --head-dtype $2 2> gpurun_out/r4_tailsets.err |
      python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('conv $1 heads $2 tail_sets $sets pairs/s', d['value'], 'ms/step', d['ms_per_step'])" >> $out || exit 1
  done
done
cat $out
