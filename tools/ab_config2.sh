#!/bin/bash
# config 2 (the small-lattice kernel) A/B of library builds in one call on one device: bash tools/ab_config2.sh ROUNDS lib1.so lib2.so ...
R=$1; shift
for i in $(seq 1 $R); do for lib in "$@"; do
FTHMC_LIB=$PWD/$lib python3 bench.py --config 2 --steps 200 --warmup 20 --regions 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 $lib', d['ms_per_step'], flush=True)"
done; done
