#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for lib in $LIBS; do for wl in c3 c3a c3b; do
  env SXG_POA_LIB=$PWD/smoothxg_amd/csrc/$lib timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-e2e --steps 1 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$wl', round(d['value'],1), 'blk/s', round(d['roofline']['kernel_ms_per_launch'],1), 'ms', d['verified'])"
done; done
