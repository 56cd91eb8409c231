#!/bin/bash
cd $GRAFT_REPO_ROOT
for w in 1 2; do PB_WPW=$w python -m pytest tests/test_hip_plan.py tests/test_hip_full.py tests/test_hip_random.py -m gpu -x -q 2>&1 | tail -2; done
bash experiments/variants3.sh r2h "4 0 12288 c2" "2 0 12288 c2" "1 0 12288 c2" "1 0 10224 c2" "1 0 8176 c2" "2 0 8176 c2" "1 0 12288 c4shard" "2 0 12288 c4shard" "4 0 7168 c3" "2 0 7168 c3" "1 0 7168 c3" "1 0 12288 c3" "4 0 7168 c1" "2 0 7168 c1" "1 0 7168 c1" "1 0 12288 c1"
export PB_LIB_PATH=$GRAFT_REPO_ROOT/experiments/libpb_abl.so
bash experiments/variants3.sh r2h_abl "4 204 12288 c2" "2 204 12288 c2" "1 204 12288 c2"
