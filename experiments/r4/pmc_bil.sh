#!/bin/bash
# usage: pmc_bil.sh <tag> <config> <counters...>: a few bilinear bench steps under rocprofv3 --pmc (kernel-trace only, own pass)
R=$GRAFT_REPO_ROOT; tag=$1; cfg=$2; shift 2; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/raw -- python3 $R/bench.py --config $cfg --sampling bilinear --steps 12 --warmup 2 --no-cpu-baseline --no-configs --no-events > $O/bench.log 2>&1
python3 $R/experiments/pmc_summary.py $O/raw | grep -A10 "bilinear.*hot" > $O/summary.txt
rm -rf $O/raw
echo "== $tag"; cat $O/summary.txt
