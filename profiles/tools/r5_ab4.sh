#!/bin/bash
# same-box A/B of two libraries on the small-block workloads (LIBS="libsxgpoa_base.so libsxgpoa.so")
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do for lib in $LIBS; do for wl in ${WLS:-c2 c2x8}; do
  env SXG_POA_LIB=$PWD/smoothxg_amd/csrc/$lib timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-e2e --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$wl', round(d['value'],1), 'blk/s', round(d['roofline']['kernel_ms_per_launch'],2), 'ms', d['verified'])"
done; done; done
