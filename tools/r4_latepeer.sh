# which stream should carry the collectives when peers can be late?  One GPU, a 1-rank RCCL
# communicator, a device-side delay of L ms in front of every gather (G = 8 steps per message) on the stream the
# collectives run on.  usage: bash tools/r4_latepeer.sh  -> gpurun_out/r4_latepeer.txt
out=gpurun_out/r4_latepeer.txt
: > $out
for late in 0 1 5 20; do
  for st in side own; do
    DODT_BENCH_COMM_STREAM=$st python3 bench.py --comm --no-cpu-baseline --no-alt --steps 200 --warmup 20 --late-peer-ms $late 2> gpurun_out/r4_latepeer.err |
      python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('late_ms $late stream $st pairs/s', d['value'], 'ms/step', d['ms_per_step'])" >> $out || exit 1
  done
done
python3 bench.py --no-cpu-baseline --no-alt --steps 200 --warmup 20 2>> gpurun_out/r4_latepeer.err | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('no exchange pairs/s', d['value'])" >> $out
cat $out
