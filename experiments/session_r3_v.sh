#!/bin/bash
# tile prologue: dims as kernel arguments + constant waves per workgroup (product) against the previous build (libpb_prev.so)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3v; mkdir -p $O
for lib in experiments/libpb_prev.so - experiments/libpb_prev.so - experiments/libpb_prev.so -; do
  timeout -k 10 300 python experiments/ab_case.py $lib c2 c1 c3 c5 c2:8 c3:8 2>> $O/ab.err | cut -c1-110 >> $O/ab.log
done
cat $O/ab.log
