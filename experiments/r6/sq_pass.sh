#!/bin/bash
# usage: sq_pass.sh <tag> <kernel-regex> <counters...> -- <program after rocprofv3 --, python3 first>
# One rocprofv3 --pmc pass (kernel-trace only, its own run; the program directly after `--`), summarised per kernel into gpurun_out/r6_sq/<tag>.txt
R=$GRAFT_REPO_ROOT; tag=$1; rx=$2; shift 2
ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
O=$R/gpurun_out/r6_sq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --pmc "${ctr[@]}" --output-format csv -d $O/raw_$tag -- "$@" > $O/$tag.log 2>&1 || { echo "FAILED $tag"; tail -5 $O/$tag.log; rm -rf $O/raw_$tag; exit 1; }
python3 $R/experiments/pmc_summary.py $O/raw_$tag | grep -A12 -E "$rx" > $O/$tag.txt
rm -rf $O/raw_$tag
echo "== $tag"; cat $O/$tag.txt
