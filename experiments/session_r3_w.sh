#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3w; mkdir -p $O
for i in 1 2 3; do
  timeout -k 10 300 python experiments/placement.py - c2 >> $O/place.log 2>> $O/err.log
  echo "-----" >> $O/place.log
done
cat $O/place.log
