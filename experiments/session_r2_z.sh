#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in c2 c5 c3 c1; do
  bash experiments/pmc.sh r2z_${cfg}_a 0 7168 $cfg TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
  bash experiments/pmc.sh r2z_${cfg}_b 0 7168 $cfg TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TA_BUSY_max
  bash experiments/pmc.sh r2z_${cfg}_c 0 7168 $cfg TCP_TA_TCP_STATE_READ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
done
