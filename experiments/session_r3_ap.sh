#!/bin/bash
# double-fisheye plans: 1 / 2 / 4 ADJACENT columns of super-tiles per XCD (wedges of one annulus share source lines)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3ap; mkdir -p $O
for run in 1 2 4 1 2 4; do
  PB_OWNER_RUN=$run timeout -k 10 300 python experiments/ab_case.py build/libphotonbend_hip_diag.so c5 c5:8 2>> $O/ab.err | cut -c24-112 | sed "s/^/OWNER_RUN=$run /" >> $O/ab.log
done
cat $O/ab.log
