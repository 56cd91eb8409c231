#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_plan.py tests/test_hip_full.py tests/test_hip_random.py tests/test_plan_api.py -m gpu -x -q 2>&1 | tail -2
bash experiments/variants.sh r2g "0 12288 c2" "0 12288 c4shard --batch 1" "0 12288 c4shard --batch 2" "0 12288 c4shard --batch 4" "0 12288 c4shard --batch 8" "0 12288 c4shard --batch 16" "0 12288 c4shard --batch 64" "0 7168 c3" "0 7168 c3 --batch 8" "0 7168 c1" "0 7168 c1 --batch 8" "0 7168 c5shard --batch 1" "0 7168 c5shard --batch 8" "0 12288 c2 --streams 2" "0 12288 c2 --streams 3"
