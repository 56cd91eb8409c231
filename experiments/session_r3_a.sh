#!/bin/bash
# round 3, session a: GPU suite on the packed-window build, then packed windows on / off (two builds) on one box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3a; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc $?" | tee -a $O/tests.log; tail -3 $O/tests.log
for rep in 1 2; do
  for lib in - experiments/libpb_nopack.so; do
    timeout -k 10 300 python experiments/ab_case.py $lib c1 c2 c3 c5 c2:8 c5:8 c3:8 >> $O/ab.log 2>> $O/ab.err || echo "ab failed: $lib" >> $O/ab.log
  done
done
timeout -k 10 300 python experiments/ab_case.py - c1@6144 c1@5120 c1@4608 c2@6144 c2@12288 c3@6144 c3@5120 c5@6144 c5@5120 c5:8@6144 >> $O/ab.log 2>> $O/ab.err
cat $O/ab.log
