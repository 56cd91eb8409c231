#!/bin/bash
# fuzz: random geometries, fast paths (windowed, direct, batch, single) against the faithful kernel byte for byte; bilinear tiles vs float64
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3y; mkdir -p $O
timeout -k 10 500 python experiments/fast_vs_faithful_sweep.py 400 20000 1 > $O/s1.log 2>&1; tail -3 $O/s1.log
timeout -k 10 500 python experiments/fast_vs_faithful_sweep.py 150 30000 5 > $O/s5.log 2>&1; tail -3 $O/s5.log
timeout -k 10 700 python experiments/fast_vs_faithful_sweep.py 120 40000 8 > $O/s8.log 2>&1; tail -3 $O/s8.log
