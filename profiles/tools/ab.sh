cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in A C; do
SXG_POA_LIB=$GRAFT_REPO_ROOT/smoothxg_amd/csrc/libsxgpoa_dev$v.so timeout 300 python bench.py --no-cpu-baseline --no-e2e --steps 2 --warmup 1 --check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['roofline']['kernel_ms_per_launch'],1), d.get('check'))"
done; done
