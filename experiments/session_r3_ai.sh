#!/bin/bash
# does the size of the rotating frame pool matter (Infinity Cache 256 MB)?
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3ai; mkdir -p $O
for mb in 320 640 1280 2560 320 2560; do
  PB_POOL_MB=$mb timeout -k 10 300 python experiments/ab_case.py - c3 c1 c2 c5 c3:8 2>> $O/ab.err | cut -c24-110 | sed "s/^/POOL_MB=$mb /" >> $O/ab.log
done
cat $O/ab.log
