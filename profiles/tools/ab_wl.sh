# same-box A/B of two builds on several workloads: profiles/tools/ab_wl.sh "c2 c3b" (libsxgpoa_devA.so vs libsxgpoa_devC.so)
cd $GRAFT_REPO_ROOT
for w in $1; do for r in 1 2; do for v in A C; do
SXG_POA_LIB=$GRAFT_REPO_ROOT/smoothxg_amd/csrc/libsxgpoa_dev$v.so timeout 400 python bench.py --workload $w --no-cpu-baseline --no-e2e --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $v', round(d['value'],1), round(d['ms_per_step'],1))"
done; done; done
