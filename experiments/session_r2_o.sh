#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in c1 c3 c5 c2; do
  for b in 6144 7168 8176 10224 12288; do
    bash experiments/variants4.sh r2o "X=1 $b $cfg"
  done
done
done
