cd $GRAFT_REPO_ROOT
for g in "" "5,2" "4,3" "4,4" "7,2" "10,1"; do
SXG_POA_FORCE_P16=$g timeout 300 python bench.py --workload c2 --no-cpu-baseline --no-e2e --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('geo [$g]', round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['kernel'], d['engine']['slots'])"
done
