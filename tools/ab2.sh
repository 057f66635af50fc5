#!/bin/bash
# A/B of library builds on config 2 (the small-lattice fused path): tools/ab2.sh ROUNDS lib1.so lib2.so ...
R=$1; shift
for i in $(seq 1 $R); do
  for lib in "$@"; do
    FTHMC_LIB=$PWD/$lib python3 bench.py --config 2 --steps 100 --warmup 10 --regions 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$lib', 'ms/step', d['ms_per_step'], 'kernel ms', r['avg_launch_ms'], 'stateless', d['stateless']['ms_per_step'], flush=True)"
  done
done
