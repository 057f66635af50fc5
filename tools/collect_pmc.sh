#!/bin/bash
# Hardware-counter passes over the bench trajectory (run on the GPU box, from the repo root):
#   bash tools/collect_pmc.sh [outdir]
# One rocprofv3 --pmc pass per counter group (TCC counters do not fit one pass together; --pmc is never
# combined with a trace domain), then tools/pmc_summary.py folds the per-dispatch CSVs into one JSON.
set -e
ROOT=$(pwd)
OUT=${1:-$ROOT/gpurun_out/pmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    echo "[pmc] pass $i: $grp" | tee -a "$OUT/progress.log"
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --regions 5 --no-cpu-baseline > "$OUT/pass$i.log" 2>&1 || echo "[pmc] pass $i failed" | tee -a "$OUT/progress.log"
done
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT" "$OUT/pmc_summary.json"
