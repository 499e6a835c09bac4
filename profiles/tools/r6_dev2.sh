#!/bin/bash
# development: which geometry does a workload run (SXG_POA_DEBUG of the built library), then single-class profile libraries
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6dev
SXG_POA_DEBUG=1 python bench.py --workload c2 --no-verify --no-cpu-baseline --no-e2e --steps 1 --warmup 0 2>&1 >/dev/null | grep -E "variant|slot time" | head -6
bash profiles/tools/r6_dev.sh "$@"
