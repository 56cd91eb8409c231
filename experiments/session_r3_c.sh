#!/bin/bash
# round 3, session c: how much of a launch is vector-ALU time?  -DPB_ABLATION, LEAN tiles with 1 instead of 4 row collapses (PB_EXP=1)
# or no polynomial at all (PB_EXP=2); c3 and c1 at a 12 KiB budget are nearly all LEAN
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3c; mkdir -p $O
for rep in 1 2; do for exp in 0 1 2; do
  PB_EXP=$exp timeout -k 10 300 python experiments/ab_case.py experiments/libpb_abl.so c3@12288 c1@12288 c3 c1 c5 c3:8@12288 2>> $O/abl.err | sed "s/^/EXP=$exp /" >> $O/abl.log
done; done
cat $O/abl.log
