#!/bin/bash
# usage: pmc_case.sh <case> <outdir> <counters...>
c=$1; out=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python3 $GRAFT_REPO_ROOT/experiments/run_case.py $c 12 > $GRAFT_REPO_ROOT/gpurun_out/$out.log 2>&1
