#!/bin/bash
# A/B of environment settings on ONE library in ONE call: tools/abenv.sh ROUNDS "ENV1=.. ENV2=.." "ENV=.." ...   ("-" = no setting)
R=$1; shift
for i in $(seq 1 $R); do
  for e in "$@"; do
    if [ "$e" = "-" ]; then ee=""; else ee="$e"; fi
    env $ee python3 bench.py --steps 30 --warmup 5 --regions 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('[$e]', 'ms/step', d['ms_per_step'], 'fwd', r['fwd_kernel_ms'], 'bwd', r['bwd_kernel_ms'], flush=True)"
  done
done
