#!/bin/bash
# Same-box A/B of the SHIPPED build under different environments: ab_envs.sh "ENV_A" "ENV_B" [bench args]
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in "$1" "$2"; do
env $v timeout 600 python bench.py --no-cpu-baseline --no-e2e --steps 2 --warmup 1 --check ${3:-} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step'],1), d['roofline']['kernel'], d['engine'])"
done; done
