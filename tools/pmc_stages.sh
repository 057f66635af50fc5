#!/bin/bash
# Instruction counts (or the counters in PMC_LIST) of the forward and the backward kernel per stage: the -DFT_DIAG build returns after stage FTHMC_DBG_STOP (1..5; 0 = whole kernel);
# differences of consecutive runs are the stages.  bash tools/pmc_stages.sh OUTDIR   (GPU box, repo root)
ROOT=$(pwd); OUT=$1; case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT";; esac; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
export FTHMC_LIB=$ROOT/experiments/lib_diag.so
for stop in 1 2 3 4 5 0; do
  FTHMC_DBG_STOP=$stop rocprofv3 --pmc ${PMC_LIST:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32} --output-format csv -d "$OUT/stop$stop" -- python3 "$ROOT/tools/kernel_loop.py" > "$OUT/stop$stop.log" 2>&1
done
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for KEY, names, stops in (('k_flow_fwd', ['stage0 plaq+sincos+weights', 'conv1', 'conv2', 'conv3', 'transform', 'finish+update'], (1, 2, 3, 4, 5, 0)),
                          ('k_flow_bwd_gather', ['stage0 loads+transform adjoint', 'conv3T', 'conv2T', 'conv1T', 'store'], (1, 2, 3, 4, 0))):
    rows = {}
    for stop in stops:
        acc = defaultdict(lambda: [0.0, 0])
        for path in glob.glob(os.path.join(out, f'stop{stop}', '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(path)):
                if KEY in r['Kernel_Name']:
                    a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
        rows[stop] = {k: v[0] / v[1] / 16384 for k, v in acc.items()}          # per wave (B=128: 2048 workgroups x 8 waves)
    prev = defaultdict(float)
    keys = sorted(rows[0])
    print(KEY + ': per wave, mean over all waves of a full-batch launch'); print('stage'.ljust(32) + ''.join(k.replace('SQ_INSTS_', '').rjust(14) for k in keys))
    for stop, nm in zip(stops, names):
        print(nm.ljust(32) + ''.join(f'{rows[stop][k] - prev[k]:14.1f}' for k in keys)); prev = rows[stop]
    print('total'.ljust(32) + ''.join(f'{rows[0][k]:14.1f}' for k in keys))
PY
