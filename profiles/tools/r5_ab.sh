#!/bin/bash
# same-box A/B of library builds (development): r5_ab.sh "<lib> <lib> ..." -- kernel ms of ns, c2 at 1000 and at 8000 blocks
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do for lib in $LIBS; do
  for wl in "ns 2" "c2 3" "c2 3 --blocks 8000"; do set -- $wl
    env SXG_POA_LIB=$PWD/smoothxg_amd/csrc/$lib timeout 600 python bench.py --workload $1 --no-cpu-baseline --no-e2e --no-verify --steps $2 --warmup 1 $3 $4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$1 $3 $4', round(d['roofline']['kernel_ms_per_launch'],2), 'ms')"
  done; done; done
if [ -n "$DBG" ]; then for a in "" "--blocks 8000"; do SXG_POA_DEBUG=1 timeout 300 python bench.py --workload c2 --no-cpu-baseline --no-e2e --no-verify --steps 1 --warmup 1 $a 2>&1 >/dev/null | grep -E "slot time|variant" | tail -2; done; fi
