#!/bin/bash
# wait-state / LDS / L2 counters per config (own passes, kernel-trace only)
cd $GRAFT_REPO_ROOT
for cfg in c5 c3 c1 c2; do
  bash experiments/pmc.sh r2y_${cfg}_a 0 7168 $cfg SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES
  bash experiments/pmc.sh r2y_${cfg}_b 0 7168 $cfg SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
  bash experiments/pmc.sh r2y_${cfg}_c 0 7168 $cfg TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum
done
