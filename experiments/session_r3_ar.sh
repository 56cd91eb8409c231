#!/bin/bash
# PB_FRONT=R: the R cheapest rows of the walk (its END: black-cornered edge rows) go first - store-only work under the launch ramp
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3ar; mkdir -p $O
for fr in 0 1 2 0 1 2; do
  PB_FRONT=$fr timeout -k 10 300 python experiments/ab_case.py build/libphotonbend_hip_diag.so c2 c1 c2:8 2>> $O/ab.err | cut -c24-112 | sed "s/^/FRONT=$fr /" >> $O/ab.log
done
cat $O/ab.log
