#!/bin/bash
# round 3, session o: the pair kernel (one wave = two tiles, 192-byte row stores) against the one-tile kernel (PB_PAIR=0), diagnostic build
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3o; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
for rep in 1 2; do for pair in 1 0; do
  PB_PAIR=$pair timeout -k 10 300 python experiments/ab_case.py build/libphotonbend_hip_diag.so c1 c2 c3 c2:8 c3:8 c1:8 2>> $O/ab.err | sed "s/^/PAIR=$pair /" >> $O/ab.log
done; done
cut -c1-125 $O/ab.log
