#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do bash experiments/variants4.sh r2x "PB_EXP=0 7168 c5" "PB_EXP=256 7168 c5" "PB_EXP=0 7168 c5shard" "PB_EXP=256 7168 c5shard"; done
