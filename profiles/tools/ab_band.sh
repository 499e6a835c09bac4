# same-box A/B of the banded workloads (libsxgpoa_devA.so vs libsxgpoa_devC.so), banded parity tests on C first
cd $GRAFT_REPO_ROOT
SXG_POA_LIB=$GRAFT_REPO_ROOT/smoothxg_amd/csrc/libsxgpoa_devC.so timeout 1500 python -m pytest tests/test_gpu_banded.py -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
for w in c3a c3b; do for v in A C; do
SXG_POA_DEBUG=1 SXG_POA_LIB=$GRAFT_REPO_ROOT/smoothxg_amd/csrc/libsxgpoa_dev$v.so timeout 600 python bench.py --workload $w --no-cpu-baseline --no-e2e --steps 2 --warmup 1 2>/tmp/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $v', round(d['value'],1), round(d['ms_per_step'],1))"
grep "round\|repeated" /tmp/err.txt | tail -2
done; done
