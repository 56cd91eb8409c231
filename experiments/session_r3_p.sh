#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3p2; mkdir -p $O
for pair in 1 0; do
  PB_PAIR=$pair timeout -k 10 300 python experiments/ab_case.py build/libphotonbend_hip_diag.so c1@4224 c1@5120 c1@7168 c1@12288 c3@5120 c3@12288 2>> $O/ab.err | sed "s/^/PAIR=$pair /" >> $O/ab.log
done
cut -c1-135 $O/ab.log
