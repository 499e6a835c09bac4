#!/bin/bash
# banded workloads: the built library against the same with the banded unit compiled for five waves per SIMD (96 VGPRs)
cd ${GRAFT_REPO_ROOT:-.}
for wl in c3b c3a; do for lib in libsxgpoa.so libsxgpoa_w5.so; do
  env SXG_POA_LIB=$PWD/smoothxg_amd/csrc/$lib python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', '$lib', round(d['value'],1), 'blk/s', round(d['ms_per_step'],1), 'ms', d['verified'], d['engine']['slots'])"
done; done
