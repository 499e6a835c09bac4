#!/bin/bash
# development: single-class libraries (see poa_kern_tables.hip.h, SXG_DEV_ONLY_W) with and without --spoa-order, debug output kept
#   r6_dev.sh "<lib>:<W,NW>:<workload>[:extra bench args]" ...
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6dev
for spec in "$@"; do IFS=: read lib force wl extra <<< "$spec"
  for o in "" "--s7-order"; do
    tag="${lib%.so}_${wl}_${o:+s7}"
    env SXG_POA_LIB=$PWD/smoothxg_amd/csrc/$lib SXG_POA_FORCE_P16="$force" SXG_POA_DEBUG=1 timeout 600 python bench.py --workload $wl $o $extra --no-verify --no-cpu-baseline --no-e2e --steps 2 --warmup 1 2> gpurun_out/r6dev/$tag.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$wl', '${o:-spoa}', round(d['value'],1), 'blk/s', round(d['ms_per_step'],1), 'ms')"
    grep -E "re-sort|slot time" gpurun_out/r6dev/$tag.err | tail -2
  done
done
