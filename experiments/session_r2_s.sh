#!/bin/bash
cd $GRAFT_REPO_ROOT
PB_ORDER=4 python -m pytest tests/test_hip_full.py tests/test_hip_plan.py -m gpu -x -q 2>&1 | tail -1
for rep in 1 2 3; do
bash experiments/variants4.sh r2s "PB_ORDER=0 7168 c2" "PB_ORDER=4 7168 c2" "PB_ORDER=0 7168 c4shard" "PB_ORDER=4 7168 c4shard"
done
