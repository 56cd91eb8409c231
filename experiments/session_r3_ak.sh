#!/bin/bash
# launch orders re-measured with COLD pools (1.25 GiB): policy (0) against the plain walk (1); and the 64 x 16 store shape (PB_EXP=1024) again
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3ak; mkdir -p $O
for ord in 0 1 0 1; do
  PB_ORDER=$ord timeout -k 10 300 python experiments/ab_case.py build/libphotonbend_hip_diag.so c1 c3 c2 c5 2>> $O/ab.err | cut -c24-112 | sed "s/^/ORDER=$ord /" >> $O/ab.log
done
for exp in 0 1024 0 1024; do
  PB_EXP=$exp timeout -k 10 300 python experiments/ab_case.py build/libphotonbend_hip_diag.so c1 c3 c1:8 c3:8 2>> $O/ab.err | cut -c24-112 | sed "s/^/EXP=$exp /" >> $O/ab.log
done
cat $O/ab.log
