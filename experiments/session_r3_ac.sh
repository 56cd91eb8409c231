#!/bin/bash
# two-eye path: coefficients read from the lanes once per eye (product) against once per row group (libpb_prev.so)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3ac; mkdir -p $O
for lib in experiments/libpb_prev.so - experiments/libpb_prev.so - experiments/libpb_prev.so -; do
  timeout -k 10 300 python experiments/ab_case.py $lib c5 c5:8 2>> $O/ab.err | cut -c1-110 >> $O/ab.log
done
cat $O/ab.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "double or full or random or bilinear" > $O/tests.log 2>&1; echo "tests rc $?"; tail -2 $O/tests.log
