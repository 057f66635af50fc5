#!/bin/bash
# The two stripe directions of the coupling kernels side by side, counters per launch (in the bench trajectory):
#   bash tools/pmc_mu.sh OUTDIR        (GPU box, repo root; one rocprofv3 --pmc pass per group, never with a trace domain)
ROOT=$(pwd); OUT=${1:-$ROOT/gpurun_out/pmcmu}; case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT";; esac
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
k=0
for g in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_avr"; do
  k=$((k+1))
  rocprofv3 --pmc $g --output-format csv -d "$OUT/pass$k" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --regions 5 --no-cpu-baseline > "$OUT/pass$k.log" 2>&1 || echo "pass $k failed ($g)"
  echo "[pmc_mu] pass $k done"
done
cd "$ROOT"; python3 tools/pmc_by_kernel.py "$OUT"/pass* | grep "k_flow_bwd_gather\|k_flow_fwd" > "$OUT/by_kernel.txt"; rm -rf "$OUT"/pass*/
