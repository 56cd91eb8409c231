#!/bin/bash
# usage: pmc_run.sh <outdir> <counters...>   (runs bench with few steps under rocprofv3 --pmc)
out=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-events > $GRAFT_REPO_ROOT/gpurun_out/$out.log 2>&1
