#!/bin/bash
# Cost map of the packed sweep's row on the small-block shape (development; see cost_map.sh): W = 10 one wave (8000 blocks)
# and W = 5 two waves (1000 blocks).  build here, run on the GPU box.
cd ${GRAFT_REPO_ROOT:-.}
MASKS="none 0 1 2 3 4 7 8 32 64 128 255"
if [ "$1" = build ]; then
  for w in 10 5; do
  for m in $MASKS; do
    X=""; [ $m != none ] && X="-DSXG_EXP=$m"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSXG_DEV_ONLY_W=$w -DSXG_DEV_ONLY_TMAX=$([ $w = 10 ] && echo 64 || echo 128) $X \
      -o smoothxg_amd/csrc/libsxgpoa_exp${w}_$m.so smoothxg_amd/csrc/sxg_poa.hip -ldl &
  done; wait; done; ls smoothxg_amd/csrc/libsxgpoa_exp*.so | wc -l
else
  for m in $MASKS; do
    env SXG_POA_LIB=$PWD/smoothxg_amd/csrc/libsxgpoa_exp10_$m.so SXG_POA_FORCE_P16=10,1 timeout 300 python bench.py --workload c2 --blocks 8000 --no-cpu-baseline --no-e2e --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EXP10x8000', '$m', round(d['roofline']['kernel_ms_per_launch'],1))"
    env SXG_POA_LIB=$PWD/smoothxg_amd/csrc/libsxgpoa_exp5_$m.so SXG_POA_FORCE_P16=5,2 timeout 300 python bench.py --workload c2 --no-cpu-baseline --no-e2e --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EXP5x1000', '$m', round(d['roofline']['kernel_ms_per_launch'],1))"
  done
fi
