#!/bin/bash
# usage: pmc.sh <outdir> <exp> <budget> <config> <counters...>   (few bench steps under rocprofv3 --pmc, kernel-trace only)
out=$1; ex=$2; bud=$3; cfg=$4; shift 4
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
PB_EXP=$ex timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out/raw -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --budget $bud --steps 12 --warmup 2 --no-cpu-baseline --no-events > $GRAFT_REPO_ROOT/gpurun_out/$out/bench.log 2>&1
python3 $GRAFT_REPO_ROOT/experiments/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/$out/raw | grep -A12 "pb_hot" > $GRAFT_REPO_ROOT/gpurun_out/$out/summary.txt
echo "== $out"; cat $GRAFT_REPO_ROOT/gpurun_out/$out/summary.txt
