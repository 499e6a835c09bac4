cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
CMD="python $R/bench.py --workload ${WL:-ns} --steps ${STEPS:-1} --warmup ${WARMUP:-1} --no-cpu-baseline --no-e2e ${EXTRA:-}"
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU" \
         "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc16_$i -o pmc -- $CMD > $OUT/pmc16_$i.log 2>&1
done
python3 - <<'PY'
import csv,glob,collections,os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out'
for d in sorted(glob.glob(out+'/pmc16_*/')):
    f=glob.glob(d+'*counter_collection.csv')
    if not f: print(d,'no csv'); continue
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        if 'poa_block' in r['Kernel_Name']: agg[r['Counter_Name']]+=float(r['Counter_Value'])
    print(d.split('/')[-2], {k:f'{v:.4g}' for k,v in agg.items()})
PY
