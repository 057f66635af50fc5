#!/bin/bash
# experiment: two chain groups out of phase (FTHMC_GROUP_LAG = cycles of a sleep kernel ahead of the second group's trajectory), per library
for lib in "$@"; do
  for lag in 0 100000 250000 500000 1000000; do
    FTHMC_LIB=$PWD/$lib FTHMC_GROUP_LAG=$lag python3 bench.py --steps 30 --warmup 5 --regions 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$lib', 'lag', $lag, 'ms/step', d['ms_per_step'], flush=True)"
  done
done
