#!/bin/bash
cd $GRAFT_REPO_ROOT
for u in 2 8; do PB_UNIT=$u python -m pytest tests/test_hip_plan.py tests/test_hip_full.py -m gpu -x -q 2>&1 | tail -1; done
for rep in 1 2 3; do
bash experiments/variants4.sh r2q "PB_UNIT=4 7168 c2" "PB_UNIT=8 7168 c2" "PB_UNIT=16 7168 c2" "PB_UNIT=2 7168 c2" "PB_UNIT=4 7168 c4shard" "PB_UNIT=8 7168 c4shard" "PB_UNIT=4 7168 c3" "PB_UNIT=8 7168 c3" "PB_UNIT=4 7168 c1" "PB_UNIT=8 7168 c1"
done
for u in 4 8 16; do PB_UNIT=$u bash experiments/pmc.sh r2q_pmc_u$u 0 7168 c2 FETCH_SIZE | grep -A1 "hot_win"; done
