#!/bin/bash
# round 3, session k: what do the generic (partly valid / wrapping) tiles cost a single launch?  PB_EXP 64 skips them, 128 skips black tiles
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3k; mkdir -p $O
for exp in 0 64 128 192; do
  PB_EXP=$exp timeout -k 10 300 python experiments/ab_case.py build/libphotonbend_hip_diag.so c2 c2:8 c1 c3 2>> $O/abl.err | sed "s/^/EXP=$exp /" >> $O/abl.log
done
cut -c1-160 $O/abl.log
