#!/bin/bash
# usage: pmc_plan.sh <tag> <config> <counters...>: plan preparation of a config under rocprofv3 --pmc (kernel-trace only, own pass)
R=$GRAFT_REPO_ROOT; tag=$1; cfg=$2; shift 2; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/raw -- python3 $R/experiments/faithful_time.py - $cfg > $O/run.log 2>&1
python3 $R/experiments/pmc_summary.py $O/raw | grep -A9 "certify\|pb_remap_kernel" > $O/summary.txt
rm -rf $O/raw
echo "== $tag"; cat $O/summary.txt
