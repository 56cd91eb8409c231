#!/bin/bash
# where the kernel arguments live: HIP_FORCE_DEV_KERNARG=0 / 1 (runtime default: unset); and streams of one process
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3aa; mkdir -p $O
for v in unset 0 1 unset 0 1; do
  if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  timeout -k 10 300 python experiments/ab_case.py - c2 c5 c1 c3 2>> $O/ab.err | cut -c1-100 | sed "s/^/KERNARG=$v /" >> $O/ab.log
done
cat $O/ab.log
