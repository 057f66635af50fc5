#!/bin/bash
# Hardware counters of the two coupling-layer kernels alone (tools/kernel_loop.py: full-batch launches, B=128, L=64):
#   bash tools/pmc_kernels.sh [outdir]     (GPU box, repo root; one rocprofv3 --pmc pass per group, no trace domain)
ROOT=$(pwd)
OUT=${1:-$ROOT/gpurun_out/pmck}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_SMEM"; do
    i=$((i+1))
    echo "[pmc] pass $i: $grp" >> "$OUT/progress.log"
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/tools/kernel_loop.py" > "$OUT/pass$i.log" 2>&1 || echo "[pmc] pass $i failed" >> "$OUT/progress.log"
done
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT" "$OUT/pmc_summary.json" > "$OUT/summary.txt" 2>&1
