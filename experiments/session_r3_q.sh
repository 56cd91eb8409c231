#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3q; mkdir -p $O
for exp in 0 1 0 1; do
  PB_EXP=$exp timeout -k 10 300 python experiments/ab_case.py build/libphotonbend_hip_diag.so c2 c2:8 2>> $O/ab.err | sed "s/^/EXP=$exp /" >> $O/ab.log
done
cut -c1-125 $O/ab.log
