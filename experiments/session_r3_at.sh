#!/bin/bash
# two-eye tiles with both eyes direct-gather: sheared gathers + LDS regrouping (product) against the row-group layout (libpb_prev.so)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3at; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "double or full or random" > $O/tests.log 2>&1; echo "tests rc $?"; tail -2 $O/tests.log
for lib in experiments/libpb_prev.so - experiments/libpb_prev.so - experiments/libpb_prev.so -; do
  timeout -k 10 300 python experiments/ab_case.py $lib c5 c5:8 2>> $O/ab.err | cut -c1-112 >> $O/ab.log
done
cat $O/ab.log
