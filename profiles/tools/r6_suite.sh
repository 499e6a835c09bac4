#!/bin/bash
# round 6: the whole GPU suite, then the headline / small-block lines in both node orders (one box)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6b
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r6b/pytest_gpu.log
tail -5 gpurun_out/r6b/pytest_gpu.log
for wl in ns c2x8 c2; do for o in "" "--s7-order"; do
  python bench.py --workload $wl $o --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/r6b/bench_${wl}_${o:+s7}.json 2> gpurun_out/r6b/bench_${wl}_${o:+s7}.err
  python -c "import json; d=json.loads(open('gpurun_out/r6b/bench_${wl}_${o:+s7}.json').read().strip().splitlines()[-1]); print('$wl', '${o:-spoa}', round(d['value'],1), 'blk/s', round(d['ms_per_step'],1), 'ms', d['verified'])"
done; done
