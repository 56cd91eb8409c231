#!/bin/bash
# store policy of single-fisheye sources with COLD frame pools (1.25 GiB): non-temporal (product) against plain (libpb_plain.so)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3aj; mkdir -p $O
for lib in - experiments/libpb_plain.so - experiments/libpb_plain.so; do
  timeout -k 10 300 python experiments/ab_case.py $lib c1 c3 c1:8 c3:8 2>> $O/ab.err | cut -c1-112 >> $O/ab.log
done
PB_POOL_MB=320 timeout -k 10 300 python experiments/ab_case.py - c1 c3 2>> $O/ab.err | cut -c1-112 | sed 's/^/POOL320 /' >> $O/ab.log
PB_POOL_MB=320 timeout -k 10 300 python experiments/ab_case.py experiments/libpb_plain.so c1 c3 2>> $O/ab.err | cut -c1-112 | sed 's/^/POOL320 /' >> $O/ab.log
cat $O/ab.log
