#!/bin/bash
cd $GRAFT_REPO_ROOT
python experiments/diag_trace.py c2 12288 > gpurun_out/trace_c2_12288.log 2>&1; cat gpurun_out/trace_c2_12288.log
python experiments/diag_trace.py c2 8176 > gpurun_out/trace_c2_8176.log 2>&1; tail -n +1 gpurun_out/trace_c2_8176.log | head -30
python experiments/diag_trace.py c3 7168 > gpurun_out/trace_c3_7168.log 2>&1; head -30 gpurun_out/trace_c3_7168.log
bash experiments/variants.sh r2e "0 12288 c2" "0 8176 c2" "0 7168 c3" "0 7168 c1"
export PB_LIB_PATH=$GRAFT_REPO_ROOT/experiments/libpb_abl.so
bash experiments/variants.sh r2e_abl "204 12288 c2" "204 8176 c2" "204 4224 c2" "48 12288 c2" "48 8176 c2"
