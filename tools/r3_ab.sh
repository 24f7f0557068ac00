# A/B of the stream set-up of bench.py (one GPU): with / without the RCCL side stream, hardware queues
cd $GRAFT_REPO_ROOT
B="timeout -k 10 200 python bench.py --no-cpu-baseline --no-alt --steps 32"
run() { tag=$1; shift; env "$@" $B $EXTRA > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/ab_$tag.json') if l.startswith('{')][-1])
print('$tag', d['value'], 'pairs/s  median step', d['step_ms']['median'], 'host', d['host_enqueue_ms_per_step'], d['config']['exchange'][-30:])
PY
}
run plain X=1 &&
EXTRA="--comm --gather-every 1" run comm_g1 X=1 &&
EXTRA="--comm --gather-every 8" run comm_g8 X=1 &&
EXTRA="--comm --gather-every 8" run comm_g8_q4 GPU_MAX_HW_QUEUES=4 &&
EXTRA="--comm --gather-every 16" run comm_g16 X=1
