#!/bin/bash
# slot placement / busy-time report of a workload with the built library (SXG_POA_DEBUG lines)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6dbg
for wl in "$@"; do
  SXG_POA_DEBUG=1 python bench.py --workload $wl --no-verify --no-cpu-baseline --no-e2e --steps 2 --warmup 1 2> gpurun_out/r6dbg/$wl.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', round(d['value'],1), 'blk/s', round(d['ms_per_step'],1), 'ms')"
  grep -E "variant|slot busy|slot time|re-sweeps" gpurun_out/r6dbg/$wl.err | tail -8
done
