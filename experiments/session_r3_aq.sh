#!/bin/bash
# VERDICT r2 item 1a built: partial vmcnt waits in the LEAN path (libpb_pwait.so, -DPB_PARTIAL_WAIT) against the single full wait (product)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3aq; mkdir -p $O
PB_LIB_PATH=$PWD/experiments/libpb_pwait.so timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "full or mid or parity or random or double or orders or plan" > $O/tests.log 2>&1; echo "tests rc $?"; tail -2 $O/tests.log
for lib in - experiments/libpb_pwait.so - experiments/libpb_pwait.so - experiments/libpb_pwait.so; do
  timeout -k 10 300 python experiments/ab_case.py $lib c3 c1 c5 c3:8 c1:8 c5:8 2>> $O/ab.err | cut -c1-112 >> $O/ab.log
done
cat $O/ab.log
