#!/bin/bash
# Copy what one call of tools/refresh_profiles.sh left in OUT (and, when given, the bench lines re-run afterwards in BENCH) into
# profiles/ under this round's names:  bash tools/install_profiles.sh ROUND OUT [BENCH]
RN=$1; R=$2; B=${3:-$2}
for f in pmc_summary pmc_summary_recomp_d1 pmc_kernels_fullbatch pmc_small; do cp $R/$f.json profiles/${RN}_$f.json; done
cp $R/kernel_stats.meta.json profiles/${RN}_kernel_stats.meta.json
for f in kernel_stats kernel_stats_config2 kernel_stats_train_L16_B512 kernel_stats_train_shard; do cp $R/$f.csv profiles/${RN}_$f.csv; done
for f in instructions_per_stage workgroup_lifetime workgroup_lifetime_train small_lattice_stage_cycles leapfrog_kernels ab_recomp_d1 train_wall pmc_train_shard ab_train_fused_vs_two_kernels; do cp $R/$f.txt profiles/${RN}_$f.txt; done
for c in 1 2 3 5; do cp $B/bench_config$c.json profiles/${RN}_bench_config$c.json; done
cp $B/bench_config2_tiled.json profiles/${RN}_bench_config2_tiled.json
if [ -f $B/bench_default.json ]; then cp $B/bench_default.json profiles/${RN}_bench.json; else cp $B/bench_config3.json profiles/${RN}_bench.json; fi
