#!/bin/bash
# Counter passes for the OTHER BASELINE shapes (c2, c3b, c4, ...): one step each (no warm-up), one counter per pass
# (--kernel-trace only, as the pool requires).  collect.py sums the counters over every poa_block dispatch of the step
# (c4 is nine launches side by side plus retry rounds) and bench.py turns them into per-cell figures.
#   WLS="c2 c3b c4 ns:nw" bash profiles/run_wl_pmc.sh      -> gpurun_out/wl_<workload>[+<mode>]_<counter>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
for ITEM in ${WLS:-c2 c3b c4}; do
  WL=${ITEM%%:*}; MODE=sw; case $ITEM in *:*) MODE=${ITEM##*:}; WL=$WL+$MODE;; esac
  CMD="python $R/bench.py --workload ${ITEM%%:*} --mode $MODE --steps 1 --warmup 0 --no-cpu-baseline --no-e2e"
  for C in SQ_INSTS_VALU FETCH_SIZE WRITE_SIZE; do
    # (the instruction pass also counts the dual-rate issues: bench.py prices full-rate opcodes issued back to back at 2 cycles)
    PMC=$C; [ $C = SQ_INSTS_VALU ] && PMC="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2"
    rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/wl_${WL}_$C -o pmc -- $CMD > $OUT/wl_${WL}_$C.log 2>&1
  done
done
