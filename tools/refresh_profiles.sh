#!/bin/bash
# Everything under profiles/ that is measured at HEAD, in one call on the GPU box (repo root):
#   bash tools/refresh_profiles.sh OUTDIR COMMIT
# bench lines of configs 3, 1, 2, 5; rocprofv3 kernel stats; the counter passes (in-bench launches and full-batch
# launches); instructions per stage (needs experiments/lib_diag.so = the -DFT_DIAG build of HEAD); workgroup lifetimes.
ROOT=$(pwd); OUT=$1; COMMIT=${2:-unknown}; mkdir -p "$OUT"; export TMPDIR=/tmp
for c in 3 1 2 5; do
  python3 bench.py --config $c --steps 100 --warmup 10 > "$OUT/bench_config$c.json" 2> "$OUT/bench_config$c.log" || echo "bench config $c failed: $?" >> "$OUT/errors.txt"
  echo "[refresh] bench config $c done"
done
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/stats.log" 2>&1)
cp $(find "$OUT/stats" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
echo "[refresh] kernel stats done"
bash tools/collect_pmc.sh "$OUT/pmc" > "$OUT/pmc.log" 2>&1
python3 tools/pmc_summary.py "$OUT/pmc" "$OUT/pmc_summary.json" "" "$COMMIT" > /dev/null
echo "[refresh] in-bench counters done"
bash tools/pmc_kernels.sh "$OUT/pmck" > "$OUT/pmck.log" 2>&1
python3 tools/pmc_summary.py "$OUT/pmck" "$OUT/pmc_kernels_fullbatch.json" "tools/kernel_loop.py: each coupling-layer kernel launched alone over the FULL batch: 128 chains x 16 tiles = 2048 workgroups of 16x16 sites (L=64, fp64), 16384 waves per launch" "$COMMIT" > /dev/null
echo "[refresh] full-batch counters done"
bash tools/pmc_stages.sh "$OUT/stg" > "$OUT/instructions_per_stage.txt" 2>&1
echo "[refresh] stages done"
python3 tools/lifetime.py 16 48 64 128 > "$OUT/workgroup_lifetime.txt" 2>&1
FTHMC_LIB=$ROOT/experiments/lib_diag.so python3 tools/lifetime.py 16 128 > "$OUT/workgroup_lifetime_diag_stamps.txt" 2>&1
rm -rf "$OUT/stats" "$OUT/pmc/pass"* "$OUT/pmck/pass"* "$OUT/stg/stop"*
echo "[refresh] done"
