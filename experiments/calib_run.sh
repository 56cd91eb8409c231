#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r2; mkdir -p $O
timeout -k 5 60 $R/experiments/exp_calib || exit 1   # plain run first: a fault must not happen under the profiler
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/raw_calib_$c -- $R/experiments/exp_calib > $O/calib_$c.log 2>&1
  python3 $R/experiments/pmc_summary.py $O/raw_calib_$c > $O/calib_pmc_$c.txt; cat $O/calib_pmc_$c.txt
done
rm -rf $O/raw_*
