#!/bin/bash
# SQ-counter summaries of the four bilinear hot kernels, pb_hot_double_kernel and pb_certify_kernel: three passes each.
# usage: sq_all.sh <suffix>   (results: gpurun_out/r6_sq/<kernel>_<suffix>_p{1,2,3}.txt)
R=$GRAFT_REPO_ROOT; S=${1:-base}; cd $R
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_CVT"
B="--steps 12 --warmup 2 --no-cpu-baseline --no-configs --no-events --no-live-traffic"
for c in ${CFGS:-c1 c2 c3 c5}; do
  i=1
  for P in "$P1" "$P2" "$P3"; do
    bash experiments/r6/sq_pass.sh ${c}_bilinear_${S}_p$i "${RX:-bilinear.*(hot|pipe)_kernel}" $P -- python3 $R/bench.py --config $c --sampling bilinear $B || exit 1
    i=$((i+1))
  done
done
if [ -z "$ONLY_BIL" ]; then
  i=1
  for P in "$P1" "$P2" "$P3"; do
    bash experiments/r6/sq_pass.sh c5_nearest_${S}_p$i "pb_hot_double_kernel" $P -- python3 $R/bench.py --config c5 $B || exit 1
    bash experiments/r6/sq_pass.sh c5_certify_${S}_p$i "certify" $P -- python3 $R/experiments/faithful_time.py - c5 || exit 1
    i=$((i+1))
  done
fi
