#!/bin/bash
cd $GRAFT_REPO_ROOT
export PB_ORDER=0
for rep in 1 2 3; do
bash experiments/variants4.sh r2l "PB_WPW=4 12288 c2" "PB_WPW=2 12288 c2" "PB_WPW=1 12288 c2" "PB_WPW=4 12288 c4shard" "PB_WPW=2 12288 c4shard" "PB_WPW=4 7168 c3" "PB_WPW=2 7168 c3" "PB_WPW=4 7168 c1" "PB_WPW=2 7168 c1" "PB_WPW=4 8176 c3" "PB_WPW=2 8176 c3" "PB_WPW=4 8176 c1" "PB_WPW=2 8176 c1"
done
