#!/bin/bash
# PB_ILV=G: frames of a batch interleaved every G workgroups inside one launch (do concurrent frames beat frame-major?)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3ah; mkdir -p $O
for g in 0 8 64 512 0 8 64 512; do
  PB_ILV=$g timeout -k 10 300 python experiments/ab_case.py build/libphotonbend_hip_diag.so c3:8 c1:8 c2:8 c3:3 c1:3 2>> $O/ab.err | cut -c24-100 | sed "s/^/ILV=$g /" >> $O/ab.log
done
cat $O/ab.log
