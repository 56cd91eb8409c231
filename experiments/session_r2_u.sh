#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_double.py tests/test_hip_full.py tests/test_hip_plan.py tests/test_plan_api.py tests/test_hip_parity.py tests/test_generic.py tests/test_hip_random.py tests/test_cli.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2 3; do bash experiments/variants4.sh r2u "X=1 7168 c5" "X=1 7168 c5shard"; done
python experiments/diag_trace.py c5_180 7168 2>&1 | grep -A8 "^LEAN"
