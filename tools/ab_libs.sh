# Same-box A/B of library builds (DODT_HIP_LIB): [BENCH_ARGS="..."] bash tools/ab_libs.sh <lib.so> [<lib.so> ...]; two rounds each
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for lib in "$@"; do
    DODT_HIP_LIB=$PWD/$lib timeout 120 python bench.py --no-cpu-baseline --no-alt --steps 40 $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-28s %7.2f pairs/s  step %.3f  frac %.4f  stacks %.3f  side-by-side %.3f' % ('$lib'.split('/')[-1], d['value'], d['step_ms']['median'], r['frac'], r['conv_stacks']['ms'], r['conv_stacks']['side_by_side_ms']))"
  done
done
