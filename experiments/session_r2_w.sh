#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for b in 6144 7168 8176 10224 12288; do bash experiments/variants4.sh r2w "X=1 $b c5" "X=1 $b c5shard"; done; done
