#!/bin/bash
# Same-box A/B of ONE build under different environments: ab_env.sh "ENV_A" "ENV_B" [lib suffix, default devC] [bench args]
cd $GRAFT_REPO_ROOT
LIB=$GRAFT_REPO_ROOT/smoothxg_amd/csrc/libsxgpoa_${3:-devC}.so
for r in 1 2; do for v in "$1" "$2"; do
env $v SXG_POA_LIB=$LIB timeout 600 python bench.py --no-cpu-baseline --no-e2e --steps 2 --warmup 1 --check ${4:-} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['roofline']['kernel_ms_per_launch'],1), d['engine'])"
done; done
