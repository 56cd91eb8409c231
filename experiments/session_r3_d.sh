#!/bin/bash
# round 3, session d: is a launch bound by its wave slots (latency x concurrency) or by a throughput?  Same tiles, same classes,
# fewer workgroups per CU: the LDS allocation of a workgroup padded (7 KiB budget = 28.7 KB per workgroup = 5 per CU;
# +4 KB -> 4 per CU, +12 KB -> 3, +25 KB -> 2, +52 KB -> 1)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3d; mkdir -p $O
for pad in 0 4096 12288 25000 53000; do
  PB_LDS_PAD=$pad timeout -k 10 300 python experiments/ab_case.py experiments/libpb_abl.so c3 c1 c2 c5 c3:8 c2:8 2>> $O/abl.err | sed "s/^/PAD=$pad /" >> $O/abl.log
done
cut -c1-120 $O/abl.log
