#!/bin/bash
# each long test of test_gpu_parity.py in its own process: which one leaves the heap damaged (abort at interpreter exit)?
cd ${GRAFT_REPO_ROOT:-.}
for k in too_long band_miss "long_sequences and 0" "long_sequences and 1" deep epochs; do
  MALLOC_CHECK_=3 timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$k" > gpurun_out/eb.log 2>&1
  echo "[$k] rc=$? $(grep -E 'passed|failed' gpurun_out/eb.log | tail -1) $(grep -E 'double free|corruption|invalid pointer|Aborted' gpurun_out/eb.log | head -2)"
done
