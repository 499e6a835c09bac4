cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_bench_contract.py 2>&1 | tail -15
for a in "" "--blocks 8000"; do timeout 300 python bench.py --workload c2 --no-cpu-baseline --no-e2e --steps 3 --warmup 1 $a 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 $a', round(d['value'],1), round(d['roofline']['kernel_ms_per_launch'],2), d['roofline']['kernel'], d['verified'])"; done
timeout 600 python bench.py --no-cpu-baseline --no-e2e --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ns', round(d['value'],1), round(d['roofline']['kernel_ms_per_launch'],2), d['roofline']['kernel'], d['verified'])"
