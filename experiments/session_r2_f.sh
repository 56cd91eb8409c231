#!/bin/bash
cd $GRAFT_REPO_ROOT
for ch in 1 2 3; do PB_CHAIN=$ch python -m pytest tests/test_hip_plan.py tests/test_hip_full.py tests/test_hip_random.py -m gpu -x -q 2>&1 | tail -2; done
bash experiments/variants2.sh r2f "1 0 12288 c2" "2 0 12288 c2" "3 0 12288 c2" "4 0 12288 c2" "8 0 12288 c2" "1 0 8176 c2" "2 0 8176 c2" "4 0 8176 c2" "1 0 7168 c3" "2 0 7168 c3" "4 0 7168 c3" "1 0 7168 c1" "2 0 7168 c1" "4 0 7168 c1" "1 0 12288 c2" "2 0 12288 c2"
python experiments/diag_trace.py c2 12288 2>&1 | grep -v "^waves\|amdgpu.ids" | head -22
