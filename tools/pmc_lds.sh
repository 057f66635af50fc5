#!/bin/bash
# LDS counters of the two coupling-layer kernels (full-batch launches) for one or more builds: tools/pmc_lds.sh [lib.so ...]
ROOT=$(pwd); export TMPDIR=/tmp
for lib in "${@:-fthmc_amd/libfthmc_hip.so}"; do
  OUT=$ROOT/gpurun_out/lds/$(basename $lib .so); rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && FTHMC_LIB=$ROOT/$lib rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/p -- python3 $ROOT/tools/kernel_loop.py > $OUT/log.txt 2>&1)
  python3 tools/pmc_summary.py $OUT $OUT/s.json > /dev/null; python3 -c "
import json
d=json.load(open('$OUT/s.json'))
for k,v in d['kernels'].items(): print('$lib', k[:16], {c:round(x['mean_per_launch']) for c,x in v.items()})"
done
