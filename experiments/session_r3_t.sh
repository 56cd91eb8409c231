#!/bin/bash
# the faithful float64 kernel and warm plan preparation: double-double functions only (libpb_ddonly.so) against the two-step evaluation (product)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3t; mkdir -p $O
for lib in experiments/libpb_ddonly.so - experiments/libpb_ddonly.so -; do
  timeout -k 10 300 python experiments/faithful_time.py $lib 2>> $O/err.log >> $O/time.log
done
cat $O/time.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "parity or mid or identity or golden or full" > $O/tests.log 2>&1; tail -3 $O/tests.log
