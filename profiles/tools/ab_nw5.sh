cd $GRAFT_REPO_ROOT
for r in 1 2; do
SXG_POA_LIB=$GRAFT_REPO_ROOT/smoothxg_amd/csrc/libsxgpoa_devA.so timeout 300 python bench.py --no-cpu-baseline --no-e2e --steps 2 --warmup 1 --check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('A', round(d['roofline']['kernel_ms_per_launch'],1), d['roofline']['kernel'], d['engine'])"
SXG_POA_FORCE_P16=9,5 SXG_POA_LIB=$GRAFT_REPO_ROOT/smoothxg_amd/csrc/libsxgpoa_devC.so timeout 300 python bench.py --no-cpu-baseline --no-e2e --steps 2 --warmup 1 --check 2>&1 | tail -3 | python -c "import json,sys; t=sys.stdin.read().strip().splitlines(); 
try:
    d=json.loads(t[-1]); print('C', round(d['roofline']['kernel_ms_per_launch'],1), d['roofline']['kernel'], d['engine'])
except Exception as e: print('C failed', t[-3:])"
done
