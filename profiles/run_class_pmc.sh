#!/bin/bash
# Which hardware counters tell full-rate VALU instructions (v_add_u32 ...: one wave instruction per ~2.4 cycles per SIMD)
# from half-rate ones (VOP3P, v_perm, DPP ...: ~4.3 cycles)?  The candidate counters are collected (a) on the opcode
# micro-benchmark profiles/ubench/op_rate.hip, where every kernel is ONE opcode, and (b) on one launch of the headline
# workload.  Output: gpurun_out/class_ubench/, gpurun_out/class_ns/ (counter_collection.csv).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 $R/profiles/ubench/op_rate.hip -o /tmp/op_rate 2> $OUT/op_rate_build.log
C="SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU_IOPS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/class_ubench -o pmc -- /tmp/op_rate > $OUT/class_ubench.log 2>&1
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/class_ns -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > $OUT/class_ns.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out'
for d in ('class_ubench', 'class_ns'):
    f = glob.glob(out + '/' + d + '/*counter_collection.csv')
    if not f:
        print(d, 'no csv'); continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'].split('(')[0][:40]
        agg.setdefault(k, collections.OrderedDict())
        agg[k][r['Counter_Name']] = agg[k].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    for k, v in agg.items():
        n = v.get('SQ_INSTS_VALU', 0) or 1
        print(d, k, ' '.join('%s=%.4g(%.3f)' % (c, x, x / n) for c, x in v.items()))
PY
