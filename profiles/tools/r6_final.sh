#!/bin/bash
# round 6, final call: the whole GPU suite first (its verdict in gpurun_out/r6_final_pytest.log), then the round's evidence set
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r6_final_pytest.log
tail -4 gpurun_out/r6_final_pytest.log
ROUND=r06 bash profiles/run_evidence.sh
