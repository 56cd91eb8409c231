#!/bin/bash
# bilinear mode, two builds alternating on one box: ab_bil.sh <tag> <libA> <libB> <config>...
R=$GRAFT_REPO_ROOT; T=$1; A=$2; B=$3; shift 3; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
for rep in 1 2 3; do for lib in $A $B; do
  PB_AB_BILINEAR=1 timeout -k 10 300 python experiments/ab_case.py $lib "$@" 2>> $O/ab.err | cut -c1-110 >> $O/ab.log
done; done
cat $O/ab.log
