#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
bash experiments/variants4.sh r2k "PB_ORDER=0 12288 c4shard" "PB_ORDER=2 12288 c4shard" "PB_ORDER=0 12288 c2" "PB_ORDER=2 12288 c2" "PB_ORDER=0 7168 c3 --batch 8" "PB_ORDER=2 7168 c3 --batch 8" "PB_ORDER=0 7168 c1 --batch 8" "PB_ORDER=2 7168 c1 --batch 8"
done
