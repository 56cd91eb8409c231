#!/bin/bash
cd $GRAFT_REPO_ROOT
./experiments/exp_req > gpurun_out/exp_req.log 2>&1; cat gpurun_out/exp_req.log
PB_EXP=3 python -m pytest tests/test_plan_api.py tests/test_hip_plan.py -m gpu -x -q 2>&1 | tail -3
