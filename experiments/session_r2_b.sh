#!/bin/bash
cd $GRAFT_REPO_ROOT
PB_EXP=3 python -m pytest tests/test_hip_plan.py tests/test_hip_full.py tests/test_hip_random.py tests/test_plan_api.py -m gpu -x -q > gpurun_out/r2b_pytest.log 2>&1; echo "pytest EXP=3 rc=$?"; tail -3 gpurun_out/r2b_pytest.log
bash experiments/variants.sh r2b "0 12288 c2" "1 12288 c2" "2 12288 c2" "3 12288 c2" "0 8176 c2" "1 8176 c2" "3 8176 c2" "3 10224 c2" "0 7168 c1" "1 7168 c1" "3 7168 c1" "3 12288 c1" "0 7168 c3" "1 7168 c3" "3 7168 c3" "3 12288 c3" "0 12288 c2" "3 12288 c2"
bash experiments/pmc.sh r2b_pmc_e0 0 12288 c2 TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE
bash experiments/pmc.sh r2b_pmc_e3 3 12288 c2 TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE
