#!/bin/bash
# phase times of the block kernel with and without the spoa order (SXG_POA_DEBUG prints the slots' phase clocks)
cd ${GRAFT_REPO_ROOT:-.}
for o in "" "--spoa-order"; do
  echo "== order: ${o:-default}"
  env SXG_POA_DEBUG=1 SXG_POA_LIB=$PWD/smoothxg_amd/csrc/$LIB SXG_POA_FORCE_P16="11,4" timeout 600 python bench.py --workload ${WL:-ns} $o --no-verify --no-cpu-baseline --no-e2e --steps 1 --warmup 0 2>&1 | grep -E "slot time|variant T" | tail -4
done
