#!/bin/bash
# same-box A/B of single-class dev libraries on the headline (forced geometry), verification on:  r6_ab.sh "<lib>:<W,NW>" ...  [REPS=2]
cd ${GRAFT_REPO_ROOT:-.}
for rep in $(seq 1 ${REPS:-2}); do for spec in "$@"; do IFS=: read lib force wl <<< "$spec"
  env SXG_POA_LIB=$PWD/smoothxg_amd/csrc/$lib SXG_POA_FORCE_P16="$force" timeout 900 python bench.py --workload ${wl:-ns} --no-cpu-baseline --no-e2e --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$force', '${wl:-ns}', round(d['value'],1), 'blk/s', round(d['roofline']['kernel_ms_per_launch'],1), 'ms', 'verified', d['verified'])"
done; done
