ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_valu; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for v in base cur; do
  if [ $v = base ]; then export FTHMC_LIB=$ROOT/experiments/lib_base.so; else unset FTHMC_LIB; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_INT32 SQ_WAVES --output-format csv -d $OUT/$v -- python3 $ROOT/tools/kernel_loop.py > $OUT/$v.log 2>&1 || echo failed $v
done
cd $ROOT
python3 - <<'PY'
import csv,glob,collections
for v in ('base','cur'):
    tot=collections.defaultdict(list)
    for f in glob.glob(f'gpurun_out/pmc_valu/{v}/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_flow_fwd' in r['Kernel_Name']:
                tot[r['Counter_Name']].append(float(r['Counter_Value']))
    w=sum(tot['SQ_WAVES'])/len(tot['SQ_WAVES'])
    print(v, {k: round(sum(x)/len(x)/w,1) for k,x in sorted(tot.items())})
PY
