#!/bin/bash
# banded sweep with the key-based end cell: banded parity, fuzz (banded modes), full-shape fixtures, then c3b / c3a
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6g
python -m pytest tests/test_gpu_banded.py tests/test_gpu_fullshape.py tests/test_gpu_fuzz.py tests/test_gpu_smooth.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r6g/pytest_band.log
tail -4 gpurun_out/r6g/pytest_band.log
for wl in c3b c3a; do
  python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/r6g/bench_${wl}.json 2> gpurun_out/r6g/bench_${wl}.err
  python -c "import json; d=json.loads(open('gpurun_out/r6g/bench_${wl}.json').read().strip().splitlines()[-1]); print('$wl', round(d['value'],1), 'blk/s', round(d['ms_per_step'],1), 'ms', d['verified'])" || tail -5 gpurun_out/r6g/bench_${wl}.err
done
