#!/bin/bash
# PB_XCD_RUN=2 / 4: runs of consecutive super-tiles of the walk on ONE XCD (half / a quarter of the seams between XCDs)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3an; mkdir -p $O
for run in 1 2 4 1 2 4; do
  PB_XCD_RUN=$run timeout -k 10 300 python experiments/ab_case.py build/libphotonbend_hip_diag.so c2 c2:8 c3 c1 c3:8 2>> $O/ab.err | cut -c24-112 | sed "s/^/RUN=$run /" >> $O/ab.log
done
cat $O/ab.log
