#!/bin/bash
# rectangular super-tiles: PB_UNIT x PB_UNIT_Y workgroups (64 px each): 4x4 = 256x256 px (product), 8x2 = 512x128, 16x1 = 1024x64, 2x8 = 128x512
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3am; mkdir -p $O
for shape in "4 0" "8 2" "16 1" "2 8" "4 0" "8 2"; do
  set -- $shape
  PB_UNIT=$1 PB_UNIT_Y=$2 timeout -k 10 300 python experiments/ab_case.py build/libphotonbend_hip_diag.so c2 c2:8 c5 c3 c1 2>> $O/ab.err | cut -c24-112 | sed "s/^/UNIT=$1x$2 /" >> $O/ab.log
done
cat $O/ab.log
