#!/bin/bash
# The bench lines that go into profiles/ AFTER tools/install_profiles.sh has put the rocprofv3 summary of the same sources there
# (bench.py then reports frac_rocprof from it, not stale):   bash tools/final_bench.sh OUTDIR     (GPU box, repo root)
OUT=${1:-gpurun_out/bf}; mkdir -p "$OUT"
for c in 3 1 2 5; do
  python3 bench.py --config $c --steps 100 --warmup 10 > "$OUT/bench_config$c.json" 2> "$OUT/bench_config$c.log" || echo "bench config $c failed: $?" >> "$OUT/errors.txt"
  echo "[final] config $c done"
done
FTHMC_SMALL_PATH=0 python3 bench.py --config 2 --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/bench_config2_tiled.json" 2> "$OUT/bench_config2_tiled.log"
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.log"      # the driver's command
echo "[final] default done"
python3 tools/run_wall.py 16 4.0 4 32 400 > "$OUT/run_wall.txt" 2>&1
python3 tools/run_wall.py 64 6.0 8 128 40 >> "$OUT/run_wall.txt" 2>&1
FTHMC_RUN_GRAPH=0 python3 tools/run_wall.py 16 4.0 4 32 400 >> "$OUT/run_wall.txt" 2>&1
echo "[final] run_wall done"
