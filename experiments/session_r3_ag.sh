#!/bin/bash
# PB_ORDER=5 / 6: heavy and light super-tiles interleaved inside ONE launch (what concurrent launches give the chip)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3ag; mkdir -p $O
for ord in 0 5 6 0 5 6; do
  PB_ORDER=$ord timeout -k 10 300 python experiments/ab_case.py build/libphotonbend_hip_diag.so c3 c1 c2 c3:8 c1:8 2>> $O/ab.err | cut -c24-100 | sed "s/^/ORDER=$ord /" >> $O/ab.log
done
cat $O/ab.log
