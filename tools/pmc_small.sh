#!/bin/bash
# Hardware counters of k_ft_small (tools/small_loop.py: config-2 trajectories, 32 workgroups of 512 threads):
#   bash tools/pmc_small.sh [outdir]     (GPU box, repo root; one rocprofv3 --pmc pass per group, no trace domain)
ROOT=$(pwd)
OUT=${1:-$ROOT/gpurun_out/pmcs}; case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT";; esac
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_SMEM"; do
    i=$((i+1))
    echo "[pmc] pass $i: $grp" >> "$OUT/progress.log"
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/tools/small_loop.py" > "$OUT/pass$i.log" 2>&1 || echo "[pmc] pass $i failed" >> "$OUT/progress.log"
done
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT" "$OUT/pmc_summary.json" "tools/small_loop.py: one launch of k_ft_small<16> = one config-2 trajectory (L=16, 4 layers, nstep 10) of 32 chains = 32 workgroups of 8 waves" "${2:-unknown}" > "$OUT/summary.txt" 2>&1
rm -rf "$OUT"/pass*/
