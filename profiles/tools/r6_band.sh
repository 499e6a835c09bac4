#!/bin/bash
# banded (-A) workloads at BASELINE's 5000 blocks: rounds of equal size against the old slot count (same box)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6d
for wl in c3b c3a; do for e in "" "SXG_POA_NO_EVEN_ROUNDS=1"; do
  env $e SXG_POA_DEBUG=1 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/r6d/bench_${wl}_${e:+old}.json 2> gpurun_out/r6d/bench_${wl}_${e:+old}.err
  python -c "import json; d=json.loads(open('gpurun_out/r6d/bench_${wl}_${e:+old}.json').read().strip().splitlines()[-1]); print('$wl', '${e:-even rounds}', round(d['value'],1), 'blk/s', round(d['ms_per_step'],1), 'ms', d['verified'], d['engine'])"
  grep -E "variant" gpurun_out/r6d/bench_${wl}_${e:+old}.err | tail -1
done; done
