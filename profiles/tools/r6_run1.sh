set -x
mkdir -p gpurun_out/r6a
python -m pytest tests/test_gpu_parity.py -x -q -k "spoa" 2>&1 | tail -15 > gpurun_out/r6a/pytest_spoa.log
tail -3 gpurun_out/r6a/pytest_spoa.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/r6a/bench_s7.json 2> gpurun_out/r6a/bench_s7.err
SXG_POA_DEBUG=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --spoa-order > gpurun_out/r6a/bench_spoa.json 2> gpurun_out/r6a/bench_spoa.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --spoa-order > gpurun_out/r6a/bench_spoa2.json 2>> gpurun_out/r6a/bench_spoa.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --workload c2x8 > gpurun_out/r6a/bench_c2x8_s7.json 2>/dev/null
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --workload c2x8 --spoa-order > gpurun_out/r6a/bench_c2x8_spoa.json 2>/dev/null
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --workload c2 > gpurun_out/r6a/bench_c2_s7.json 2>/dev/null
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --workload c2 --spoa-order > gpurun_out/r6a/bench_c2_spoa.json 2>/dev/null
