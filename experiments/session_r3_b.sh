#!/bin/bash
# round 3, session b: where does a launch's time go?  -DPB_ABLATION build, PB_EXP bits: 4 skip DIRECT (incl. packed), 8 skip LEAN,
# 64 skip generic, 128 skip BLACK, 16 no stores, 32 no loads
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3b; mkdir -p $O
for exp in 0 16 32 48 4 8 12 204; do
  PB_EXP=$exp timeout -k 10 300 python experiments/ab_case.py experiments/libpb_abl.so c1 c3 c2 c5 c3:8 2>> $O/abl.err | sed "s/^/EXP=$exp /" >> $O/abl.log
done
cat $O/abl.log
