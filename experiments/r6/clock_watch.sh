#!/bin/bash
# clock_watch.sh <label> <lib or ""> <case>: the probe for ~3 s of launches with rocm-smi sampled beside it (sclk, power) - is a build clock- or power-limited on this box?
R=$GRAFT_REPO_ROOT; cd $R; label=$1; lib=$2; c=$3
if [ -n "$lib" ]; then export PB_LIB_PATH=$R/$lib; else unset PB_LIB_PATH; fi
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Socket Power" | tr -s ' \t' ' ' | tr '\n' ';'; echo; sleep 0.1; done ) > /tmp/smi_$label.txt &
W=$!
python3 experiments/r6/c5_bil_probe.py $c --json --bil-only --reps 3000 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["us_min_p10_med"]["bilinear"], d["shape"])' $label
wait $W
echo "  smi samples (sclk MHz ; W):"; sort /tmp/smi_$label.txt | uniq -c | sort -rn | head -6
