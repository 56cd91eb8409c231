#!/bin/bash
# two-step (Ziv) evaluation with the double-double paths inlined (libpb_ziv_inline.so) against real calls (product)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3u; mkdir -p $O
for lib in experiments/libpb_ziv_inline.so - experiments/libpb_ziv_inline.so -; do
  timeout -k 10 300 python experiments/faithful_time.py $lib 2>> $O/err.log >> $O/time.log
done
cat $O/time.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "parity or mid or identity or golden or full" > $O/tests.log 2>&1; tail -3 $O/tests.log
