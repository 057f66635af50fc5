#!/bin/bash
# A/B/C... of several builds of libfthmc_hip.so on ONE device in ONE call: tools/abn.sh ROUNDS lib1.so lib2.so ...
R=$1; shift
for i in $(seq 1 $R); do
  for lib in "$@"; do
    FTHMC_LIB=$PWD/$lib python3 bench.py --steps 30 --warmup 5 --regions 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$lib', 'ms/step', d['ms_per_step'], 'fwd', r['fwd_kernel_ms'], 'bwd', r['bwd_kernel_ms'], 'full fwd', r['full_batch_exclusive']['fwd_kernel_ms'], 'bwd', r['full_batch_exclusive']['bwd_kernel_ms'], flush=True)"
  done
done
