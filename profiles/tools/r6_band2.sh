#!/bin/bash
# 2-byte band cells: the banded parity tests, the full-shape banded fixtures, then c3b / c3a against the 4-byte cells (same box)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6f
python -m pytest tests/test_gpu_banded.py tests/test_gpu_fullshape.py -m gpu -x -q -k "banded or band or c3b or adaptive or static" 2>&1 | tail -15 > gpurun_out/r6f/pytest_band.log
tail -4 gpurun_out/r6f/pytest_band.log
for wl in c3b c3a; do for e in "" "SXG_POA_BAND_CELL_BYTES=4"; do
  env $e python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/r6f/bench_${wl}_${e:+cb4}.json 2> gpurun_out/r6f/bench_${wl}_${e:+cb4}.err
  python -c "import json; d=json.loads(open('gpurun_out/r6f/bench_${wl}_${e:+cb4}.json').read().strip().splitlines()[-1]); print('$wl', '${e:-2-byte cells}', round(d['value'],1), 'blk/s', round(d['ms_per_step'],1), 'ms', d['verified'], d['engine']['slots'], d['engine']['arena_bytes'])" || tail -5 gpurun_out/r6f/bench_${wl}_${e:+cb4}.err
done; done
