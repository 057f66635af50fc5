#!/bin/bash
# A/B of two builds of libfthmc_hip.so on ONE device in ONE call: alternates bench.py runs (config 3, no CPU leg).
# usage: tools/ab.sh [rounds] [extra bench args]   (A = experiments/lib_base.so, B = fthmc_amd/libfthmc_hip.so)
R=${1:-2}; shift
mkdir -p gpurun_out/r2
for i in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then export FTHMC_LIB=$PWD/experiments/lib_base.so; else unset FTHMC_LIB; fi
    python3 bench.py --steps 30 --warmup 5 --regions 5 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', 'ms/step', d['ms_per_step'], 'fwd', r['fwd_kernel_ms'], 'bwd', r['bwd_kernel_ms'], 'full fwd', r['full_batch_exclusive']['fwd_kernel_ms'], 'bwd', r['full_batch_exclusive']['bwd_kernel_ms'], flush=True)"
  done
done
