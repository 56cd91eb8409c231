#!/bin/bash
cd $GRAFT_REPO_ROOT
export PB_LIB_PATH=$GRAFT_REPO_ROOT/experiments/libpb_abl.so
bash experiments/variants.sh r2d "0 12288 c2" "4 12288 c2" "8 12288 c2" "12 12288 c2" "16 12288 c2" "32 12288 c2" "48 12288 c2" "64 12288 c2" "128 12288 c2" "76 12288 c2" "204 12288 c2" "0 4224 c2" "16 4224 c2" "32 4224 c2" "0 12288 c3" "4 12288 c3" "8 12288 c3" "16 12288 c3" "32 12288 c3" "0 7168 c3" "16 7168 c3" "32 7168 c3"
