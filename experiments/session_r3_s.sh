#!/bin/bash
# PB_EXP=1024: the hot kernel's stores in the shape of a 64 x 16 tile (whole 192-byte row pieces per instruction), loads unchanged
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3s; mkdir -p $O
for exp in 0 1024 0 1024; do
  PB_EXP=$exp timeout -k 10 300 python experiments/ab_case.py build/libphotonbend_hip_diag.so c3 c1 c3:8 c1:8 c2 2>> $O/ab.err | sed "s/^/EXP=$exp /" >> $O/ab.log
done
cut -c1-125 $O/ab.log
