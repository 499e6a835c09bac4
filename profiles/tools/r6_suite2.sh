#!/bin/bash
# round 6: the tests touched since the last full suite, then the default bench line as the driver runs it
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6c
python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r6c/pytest_gpu.log
tail -6 gpurun_out/r6c/pytest_gpu.log
python bench.py > gpurun_out/r6c/bench_default.json 2> gpurun_out/r6c/bench_default.err
python -c "
import json; d=json.loads(open('gpurun_out/r6c/bench_default.json').read().strip().splitlines()[-1])
print(round(d['value'],1), 'blk/s', round(d['ms_per_step'],1), 'ms', d['verified'], d['config']['order'][:12], 'copy', d['roofline']['hbm']['copy_GBps_measured'], 'roctx', d['engine']['roctx_ranges'], 'e2e', d.get('end_to_end',{}).get('ratio_to_kernel_only'), 'cpu', d['cpu_baseline']['value'], d['roofline']['valu_frac'], d['roofline']['algorithmic_frac'])"
for wl in c2x8 c2; do python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r6c/bench_$wl.json 2>/dev/null;  python -c "
import json; d=json.loads(open('gpurun_out/r6c/bench_$wl.json').read().strip().splitlines()[-1])
print('$wl', round(d['value'],1), 'blk/s', round(d['ms_per_step'],1), 'ms', d['verified'], 'e2e', d.get('end_to_end',{}).get('ratio_to_kernel_only'))"; done
