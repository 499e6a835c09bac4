#!/bin/bash
# price of spoa's node order (S7', the default since round 6): the same workload in the default order and with --s7-order (the
# incrementally kept order of rounds 1-5), same box.
#   LIBS="libsxgpoa.so" WLS="ns c2x8" bash profiles/tools/s7_ab.sh            (the built library)
#   LIBS="a.so b.so" FORCE="11,4" bash profiles/tools/s7_ab.sh                 (dev libraries of ONE packed class, side by side)
cd ${GRAFT_REPO_ROOT:-.}
[ -n "$FORCE" ] && export SXG_POA_FORCE_P16="$FORCE"
for lib in $LIBS; do for wl in ${WLS:-ns}; do for o in "" "--s7-order"; do
  env SXG_POA_LIB=$PWD/smoothxg_amd/csrc/$lib timeout 600 python bench.py --workload $wl $o --no-verify --no-cpu-baseline --no-e2e --steps 1 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$wl', '${o:-spoa (default)}', round(d['value'],1), 'blk/s', round(d['ms_per_step'],1), 'ms')"
done; done; done
