#!/bin/bash
# the same prologue hint in the double-fisheye kernel (product) against the build before both prologue changes (libpb_prev.so); then the suite
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3x; mkdir -p $O
for lib in experiments/libpb_prev.so - experiments/libpb_prev.so - experiments/libpb_prev.so -; do
  timeout -k 10 300 python experiments/ab_case.py $lib c5 c5:8 c1 c3 2>> $O/ab.err | cut -c1-110 >> $O/ab.log
done
cat $O/ab.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -2 $O/tests.log
