#!/bin/bash
# A/B of library builds on config 2: ab_c2.sh ROUNDS lib1 lib2 ...
R=$1; shift
for i in $(seq 1 $R); do
  for lib in "$@"; do
    FTHMC_LIB=$PWD/$lib python3 bench.py --config 2 --steps 200 --warmup 20 --regions 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'config 2 ms/step', d['ms_per_step'], flush=True)"
  done
done
