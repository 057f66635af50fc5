#!/bin/bash
# Everything under profiles/ that is measured at HEAD, in one call on the GPU box (repo root):
#   bash tools/refresh_profiles.sh OUTDIR COMMIT
# bench lines of configs 3, 1, 2, 5 (+ config 2 on the tiled path); rocprofv3 kernel stats of the headline and of config 2;
# the counter passes (in-bench launches, full-batch launches, the small-lattice kernel, the plain-HMC leapfrog kernels);
# instructions per stage (needs experiments/lib_diag.so = the -DFT_DIAG build of HEAD); workgroup lifetimes; the A/B of the
# act'(z1)-recompute build (experiments/lib_recomp_d1.so = make -C fthmc_amd/csrc recomp) with its HBM traffic; round 6: the fused
# training backward's stage cycles and its A/B against the two-kernel form (lib_bt_stamps.so, lib_twokernel.so: tools/build_variant.sh).
ROOT=$(pwd); OUT=$1; COMMIT=${2:-unknown}; mkdir -p "$OUT"; export TMPDIR=/tmp
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT";; esac
for c in 3 1 2 5; do
  python3 bench.py --config $c --steps 100 --warmup 10 > "$OUT/bench_config$c.json" 2> "$OUT/bench_config$c.log" || echo "bench config $c failed: $?" >> "$OUT/errors.txt"
  echo "[refresh] bench config $c done"
done
FTHMC_SMALL_PATH=0 python3 bench.py --config 2 --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/bench_config2_tiled.json" 2> "$OUT/bench_config2_tiled.log"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --regions 5 --no-cpu-baseline > "$OUT/stats.log" 2>&1)
cp $(find "$OUT/stats" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
# which kernels the summary was taken on (bench.py: roofline.rocprof_source)
python3 -c "import json,sys; sys.path.insert(0,'tools'); from csrc_sha import csrc_sha16; json.dump({'csrc_sha16': csrc_sha16('.'), 'commit': '$COMMIT', 'command': 'rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --regions 5 --no-cpu-baseline'}, open('$OUT/kernel_stats.meta.json','w'))"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats2" -- python3 "$ROOT/bench.py" --config 2 --steps 20 --warmup 2 --regions 5 --no-cpu-baseline > "$OUT/stats2.log" 2>&1)
cp $(find "$OUT/stats2" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats_config2.csv"
echo "[refresh] kernel stats done"
bash tools/collect_pmc.sh "$OUT/pmc" > "$OUT/pmc.log" 2>&1
python3 tools/pmc_summary.py "$OUT/pmc" "$OUT/pmc_summary.json" "" "$COMMIT" > /dev/null
echo "[refresh] in-bench counters done"
FTHMC_LIB=$ROOT/experiments/lib_recomp_d1.so bash tools/collect_pmc.sh "$OUT/pmcr" > "$OUT/pmcr.log" 2>&1
python3 tools/pmc_summary.py "$OUT/pmcr" "$OUT/pmc_summary_recomp_d1.json" "bench.py config 3, two chain groups (64-chain launches), library built with -DFT_RECOMP_D1=1: act'(z1) recomputed in the backward, not stashed" "$COMMIT" > /dev/null
echo "[refresh] recompute-build traffic done"
bash tools/pmc_kernels.sh "$OUT/pmck" > "$OUT/pmck.log" 2>&1
python3 tools/pmc_summary.py "$OUT/pmck" "$OUT/pmc_kernels_fullbatch.json" "tools/kernel_loop.py: each coupling-layer kernel launched alone over the FULL batch: 128 chains x 16 tiles = 2048 workgroups of 16x16 sites (L=64, fp64), 16384 waves per launch; k_leap_rows / k_force<1>: one plain-HMC leapfrog step of 128 chains" "$COMMIT" > /dev/null
echo "[refresh] full-batch counters done"
bash tools/pmc_small.sh "$OUT/pmcs" "$COMMIT" > "$OUT/pmcs.log" 2>&1
cp "$OUT/pmcs/pmc_summary.json" "$OUT/pmc_small.json"
echo "[refresh] small-lattice counters done"
bash tools/pmc_stages.sh "$OUT/stg" > "$OUT/instructions_per_stage.txt" 2>&1
echo "[refresh] stages done"
python3 tools/lifetime.py 16 48 64 128 > "$OUT/workgroup_lifetime.txt" 2>&1
python3 tools/small_profile.py > "$OUT/small_lattice_stage_cycles.txt" 2>&1
python3 tools/leap_check.py > "$OUT/leapfrog_kernels.txt" 2>&1
bash tools/abn.sh 2 fthmc_amd/libfthmc_hip.so experiments/lib_recomp_d1.so > "$OUT/ab_recomp_d1.txt" 2>&1
# round 4: whole training steps (wall) against the compute-only figure, and where a step's GPU time goes at L = 16, batch 512
python3 tools/train_wall.py 8 512 8 200 16 512 8 200 16 64 8 200 12 128 8 200 256 32 16 10 > "$OUT/train_wall.txt" 2> "$OUT/train_wall.err"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats3" -- python3 "$ROOT/tools/train_trace.py" 16 512 8 50 > "$OUT/stats3.log" 2>&1)
cp $(find "$OUT/stats3" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats_train_L16_B512.csv"
# the training gradient at the config-5 shard on one stream: kernel stats and counters per kernel (tools/pmc_by_kernel.py)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats4" -- python3 "$ROOT/tools/train_trace.py" 256 32 16 5 1 > "$OUT/stats4.log" 2>&1)
cp $(find "$OUT/stats4" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats_train_shard.csv"
k=0
for g in "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_LDS"; do
  k=$((k+1))
  (cd /tmp && rocprofv3 --pmc $g --output-format csv -d "$OUT/pmct/pass$k" -- python3 "$ROOT/tools/train_trace.py" 256 32 16 2 1 > "$OUT/pmct$k.log" 2>&1)
done
python3 tools/pmc_by_kernel.py "$OUT"/pmct/pass* > "$OUT/pmc_train_shard.txt"
python3 tools/lifetime.py 512 2>&1 | grep -v "flow_fwd\|flow_bwd:" > "$OUT/workgroup_lifetime_train.txt"
# round 6: the fused training backward (flow_bwd_train.hip): cycles per stage and item (experiments/lib_bt_stamps.so = tools/build_variant.sh bt_stamps
# -DFT_BT_STAMPS at HEAD) and the alternating A/B against the two-kernel form (experiments/lib_twokernel.so = -DFT_FUSED_WGRAD=0 at HEAD)
FTHMC_LIB=$ROOT/experiments/lib_bt_stamps.so python3 tools/bt_stamp_run.py 2>&1 | grep "bwd_train wg" >> "$OUT/workgroup_lifetime_train.txt"
bash tools/ab_train.sh 2 fthmc_amd/libfthmc_hip.so experiments/lib_twokernel.so > "$OUT/ab_train_fused_vs_two_kernels.txt" 2>&1
echo "[refresh] training done"
rm -rf "$OUT/stats" "$OUT/stats2" "$OUT/stats3" "$OUT/stats4" "$OUT"/pmc*/pass*/ "$OUT/stg/stop"*
echo "[refresh] done"
