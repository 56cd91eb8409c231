#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in 1 2 3; do PB_DOUBLE_FPW=$f python -m pytest tests/test_hip_double.py tests/test_hip_plan.py tests/test_hip_full.py -m gpu -x -q 2>&1 | tail -2; done
for f in 1 2 4 8; do export PB_DOUBLE_FPW=$f; echo "fpw $f"; bash experiments/variants.sh r2i_$f "0 7168 c5shard --batch 8" "0 7168 c5shard --batch 1" "0 8176 c5shard --batch 16" "0 12288 c5shard --batch 8"; done
