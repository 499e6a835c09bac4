#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/c2diag; mkdir -p $O
run() { n=$1; shift
  env "$@" SXG_POA_DEBUG=1 timeout 300 python bench.py --workload c2 --no-cpu-baseline --no-e2e --steps 3 --warmup 1 $ARGS > $O/$n.json 2> $O/$n.err
  echo "== $n"; grep -E "variant|slot time|slot busy|row profile|traceback:" $O/$n.err | tail -5
}
ARGS="" run ns1000_10_1 SXG_POA_NO_SPREAD=1
ARGS="" run nsp1000_10_1 SXG_POA_NO_SPREAD=1 SXG_POA_LIB=$PWD/smoothxg_amd/csrc/libsxgpoa_prof_10.so
ARGS="--blocks 2000" run d2000 X=1
ARGS="--blocks 2000" run ns2000 SXG_POA_NO_SPREAD=1
ARGS="--blocks 16000" run d16000 X=1
