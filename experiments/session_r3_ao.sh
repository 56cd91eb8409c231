#!/bin/bash
# the three walks forced (PB_ORDER=1 plain, 2 rows from the heaviest outwards, 3 super-tiles heaviest first) against the policy (0), COLD pools
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3ao; mkdir -p $O
for ord in 0 1 2 3 0 1 2 3; do
  PB_ORDER=$ord timeout -k 10 300 python experiments/ab_case.py build/libphotonbend_hip_diag.so c1 c3 c2 c1:8 c3:8 c2:8 2>> $O/ab.err | cut -c24-112 | sed "s/^/ORDER=$ord /" >> $O/ab.log
done
cat $O/ab.log
