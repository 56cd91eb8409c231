#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
bash experiments/variants4.sh r2n "PB_EXP=0 12288 c2" "PB_EXP=2 12288 c2" "PB_EXP=0 12288 c4shard" "PB_EXP=2 12288 c4shard"
done
bash experiments/pmc.sh r2n_pmc_e0 0 12288 c2 WRITE_SIZE
bash experiments/pmc.sh r2n_pmc_e2 2 12288 c2 WRITE_SIZE
