#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then PMC passes (each in its own run,
# --kernel-trace only, as the pool requires).  Results land in gpurun_out/prof_*.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
CMD="python $R/bench.py --workload ${WL:-ns} --blocks ${BLOCKS:-512} --steps 1 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o stats -- $CMD > $OUT/bench_stats.log 2>&1
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" \
         "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C -d $OUT/prof_pmc_$tag -o pmc -- $CMD > $OUT/bench_pmc_$tag.log 2>&1
done
ls -R $OUT | head -80
