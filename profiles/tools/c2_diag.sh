#!/bin/bash
# c2 diagnostics (development): slot-time split and row profile of the small-block shape at 1000 and 8000 blocks.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/c2diag; mkdir -p $O
run() { # name, env..., -- args
  n=$1; shift
  env "$@" SXG_POA_DEBUG=1 timeout 300 python bench.py --workload c2 --no-cpu-baseline --no-e2e --steps 3 --warmup 1 $ARGS > $O/$n.json 2> $O/$n.err
  python - <<P
import json
try:
    d=json.loads(open("$O/$n.json").read().strip().splitlines()[-1]); print("$n", round(d['value'],1), "blk/s", round(d['ms_per_step'],2), "ms", d['roofline']['kernel'], d['engine']['slots'])
except Exception as e: print("$n failed", e)
P
  grep -E "variant|slot time|slot busy|row profile|traceback:" $O/$n.err | tail -8
}
ARGS="" run d1000 X=1
ARGS="" run f1000_10_1 SXG_POA_FORCE_P16=10,1
ARGS="--blocks 8000" run d8000 X=1
ARGS="--blocks 4000" run d4000 X=1
ARGS="" run p1000_5_2 SXG_POA_LIB=$PWD/smoothxg_amd/csrc/libsxgpoa_prof_5.so SXG_POA_FORCE_P16=5,2
ARGS="" run p1000_10_1 SXG_POA_LIB=$PWD/smoothxg_amd/csrc/libsxgpoa_prof_10.so SXG_POA_FORCE_P16=10,1
ARGS="--blocks 8000" run p8000_10_1 SXG_POA_LIB=$PWD/smoothxg_amd/csrc/libsxgpoa_prof_10.so SXG_POA_FORCE_P16=10,1
