#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
bash experiments/variants4.sh r2m "PB_ORDER=0 12288 c2" "PB_ORDER=3 12288 c2" "PB_ORDER=0 12288 c4shard" "PB_ORDER=3 12288 c4shard"
done
