#!/bin/bash
# Profiles bench.py on the GPU box.  Kernel trace + stats first; hardware counters each in their
# own run (--kernel-trace only, as the pool requires).  Output: gpurun_out/prof_*/ (CSV).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
CMD="python $R/bench.py --workload ${WL:-ns} --steps ${STEPS:-1} --warmup ${WARMUP:-1} --no-cpu-baseline --no-e2e ${EXTRA:-}"
[ -z "$NOSTATS" ] && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o stats -- $CMD > $OUT/bench_stats.log 2>&1
for C in ${COUNTERS:-FETCH_SIZE WRITE_SIZE}; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/prof_pmc_$C -o pmc -- $CMD > $OUT/bench_pmc_$C.log 2>&1
done
find $OUT -name "*.csv" | head -40
