#!/bin/bash
# round 3, session f: c5 - what do the two-eye tiles cost (PB_EXP 256 skips them, 512 skips the one-eye tiles, 768 both = the launch alone)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3f; mkdir -p $O
for exp in 0 256 512 768 272 528; do
  PB_EXP=$exp timeout -k 10 300 python experiments/ab_case.py experiments/libpb_abl.so c5 c5:8 2>> $O/abl.err | sed "s/^/EXP=$exp /" >> $O/abl.log
done
cut -c1-120 $O/abl.log
