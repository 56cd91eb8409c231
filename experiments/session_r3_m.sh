#!/bin/bash
# round 3, session m: stagger the first round of workgroups (wave w of a workgroup sleeps w * u * 64 clocks before its entry load)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3m; mkdir -p $O
for u in 0 10 20 40 63; do
  exp=$((1024 + u * 4096))
  PB_EXP=$exp timeout -k 10 300 python experiments/ab_case.py build/libphotonbend_hip_diag.so c2 c3 c1 2>> $O/abl.err | sed "s/^/U=$u /" >> $O/abl.log
done
cut -c1-120 $O/abl.log
