import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d["roofline"]
print("pairs/s", d["value"], "ms", d["ms_per_step"], "stacks alone", r["conv_stacks"]["ms"], "side by side", r["conv_stacks"]["side_by_side_ms"])
print(' '.join('%s:%s:%.0f' % (l['net'][0], l['name'].replace('pyramid_','').replace('conv','c'), l['us'] or 0) for l in r['layers']))
