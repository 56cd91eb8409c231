#!/bin/bash
# round 3, session g: the parameter block by value (1216-byte kernel argument segment) against by pointer (plan-resident device
# copy, 120 bytes of arguments): whole launches and the all-tiles-skipped launch (PB_EXP=204 / 768)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3g; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_hip_plan.py tests/test_hip_full.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2 3; do
  for lib in - experiments/libpb_pptr.so; do
    timeout -k 10 300 python experiments/ab_case.py $lib c2 c1 c3 c5 c2:8 2>> $O/ab.err >> $O/ab.log
  done
  for lib in experiments/libpb_abl.so experiments/libpb_abl_pptr.so; do
    PB_EXP=204 timeout -k 10 300 python experiments/ab_case.py $lib c2 c1 2>> $O/ab.err | sed "s/^/SKIPALL /" >> $O/ab.log
    PB_EXP=768 timeout -k 10 300 python experiments/ab_case.py $lib c5 2>> $O/ab.err | sed "s/^/SKIPALL /" >> $O/ab.log
  done
done
cut -c1-110 $O/ab.log
