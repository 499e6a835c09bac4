#!/bin/bash
# Cost map of the packed sweep's row (development): builds the headline class with a SECOND, throw-away sweep in front of
# every real one (-DSXG_EXP=<mask>, see dp_fill_p16) and times the bench; the difference between two masks prices the
# parts of the row one of them leaves out.  Built HERE (no GPU needed): profiles/tools/cost_map.sh build ; run on the
# GPU box: profiles/tools/cost_map.sh run
cd ${GRAFT_REPO_ROOT:-.}
MASKS="none 0 1 2 4 8 16 32 64 128 255"
if [ "$1" = build ]; then
  for m in $MASKS; do
    X=""; [ $m != none ] && X="-DSXG_EXP=$m"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSXG_DEV_ONLY_W=11 -DSXG_DEV_ONLY_TMAX=256 $X \
      -o smoothxg_amd/csrc/libsxgpoa_exp_$m.so smoothxg_amd/csrc/sxg_poa.hip -ldl &
  done; wait; ls -la smoothxg_amd/csrc/libsxgpoa_exp_*.so
else
  for m in $MASKS; do
    env SXG_POA_LIB=$PWD/smoothxg_amd/csrc/libsxgpoa_exp_$m.so SXG_POA_FORCE_P16=11,4 SXG_POA_MERGE=0.5 timeout 900 python bench.py --no-cpu-baseline --no-e2e --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EXP', '$m', round(d['ms_per_step'],1))"
  done
fi
