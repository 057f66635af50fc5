for i in 1 2 3; do for lib in experiments/lib_base.so fthmc_amd/libfthmc_hip.so; do
FTHMC_LIB=$PWD/$lib python3 bench.py --config 2 --steps 200 --warmup 20 --regions 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 $lib', d['ms_per_step'], flush=True)"
done; done
