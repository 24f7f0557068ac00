# Same-box A/B of one environment switch: bash tools/ab_env.sh VAR A B [bench.py args]  (two rounds each, 40 steps)
cd $GRAFT_REPO_ROOT
var=$1; va=$2; vb=$3; shift 3
for round in 1 2; do
  for val in "$va" "$vb"; do
    env "$var=$val" timeout 120 python bench.py --no-cpu-baseline --no-alt --steps 40 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-28s %7.2f pairs/s  step %.3f  frac %.4f  stacks %.3f  side-by-side %.3f' % ('$var=$val', d['value'], d['step_ms']['median'], r['frac'], r['conv_stacks']['ms'], r['conv_stacks']['side_by_side_ms']))"
  done
done
