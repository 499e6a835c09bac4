#!/bin/bash
# round 6: the whole GPU suite, then the main workloads in the default order (one box)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6e
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r6e/pytest_gpu.log
tail -5 gpurun_out/r6e/pytest_gpu.log
for wl in ns c2x8 c2 c3b c3a c3 "ns --mode nw" "ns4 --mode nw"; do set -- $wl
  python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/r6e/bench_$1$3.json 2> gpurun_out/r6e/bench_$1$3.err
  python -c "import json; d=json.loads(open('gpurun_out/r6e/bench_$1$3.json').read().strip().splitlines()[-1]); print('$wl', round(d['value'],1), 'blk/s', round(d['roofline']['kernel_ms_per_launch'],1), 'ms', d['verified'])"
done
