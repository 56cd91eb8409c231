#!/bin/bash
# round 3, session e: single launches against batches with and without stores (is the single-launch penalty of the
# plain-store configs the end-of-kernel write-back of dirty L2 lines?)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3e; mkdir -p $O
for exp in 0 16 32; do
  PB_EXP=$exp timeout -k 10 300 python experiments/ab_case.py experiments/libpb_abl.so c2 c2:2 c2:8 c5 c5:8 c3 c3:8 c1 c1:8 2>> $O/abl.err | sed "s/^/EXP=$exp /" >> $O/abl.log
done
cut -c1-120 $O/abl.log
