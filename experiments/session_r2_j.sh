#!/bin/bash
cd $GRAFT_REPO_ROOT
for o in 0 1; do PB_ORDER=$o python -m pytest tests/test_hip_plan.py tests/test_hip_full.py tests/test_hip_random.py tests/test_plan_api.py tests/test_hip_mid.py -m gpu -x -q 2>&1 | tail -2; done
for rep in 1 2; do
bash experiments/variants4.sh r2j "PB_ORDER=0 12288 c2" "PB_ORDER=1 12288 c2" "PB_ORDER=2 12288 c2" "PB_ORDER=0 7168 c3" "PB_ORDER=1 7168 c3" "PB_ORDER=2 7168 c3" "PB_ORDER=0 7168 c1" "PB_ORDER=1 7168 c1" "PB_ORDER=2 7168 c1" "PB_ORDER=0 12288 c4shard" "PB_ORDER=1 12288 c4shard"
done
python experiments/diag_trace.py c2 12288 2>&1 | grep "alive\|span"
