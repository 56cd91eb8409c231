#!/bin/bash
# round 3, session l: the MASKED tile class (ring tiles of a fisheye destination through the direct-gather path): whole suite, timings
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3l; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
for rep in 1 2; do
timeout -k 10 300 python experiments/ab_case.py - c2 c2:8 c1 c3 c5 2>> $O/ab.err >> $O/ab.log
done
cut -c1-175 $O/ab.log
