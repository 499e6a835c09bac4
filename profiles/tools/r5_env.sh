#!/bin/bash
# same-box runs of one library under different environments (development): ENVS="A=1|B=2 C=3|..." WL="c2 3 --blocks 8000"
cd ${GRAFT_REPO_ROOT:-.}
IFS='|' read -ra EV <<< "$ENVS"
for rep in 1 2; do for ev in "${EV[@]}"; do
  env $ev timeout 600 python bench.py --workload $WL --no-cpu-baseline --no-e2e --no-verify --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$ev]', '$WL', round(d['roofline']['kernel_ms_per_launch'],2), 'ms', d['roofline']['kernel'], d['engine']['slots'])"
done; done
