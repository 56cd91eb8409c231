#!/bin/bash
# bilinear kernel time per config and window budget ($1 = tag)
R=$GRAFT_REPO_ROOT; T=$1; shift; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
for c in "$@"; do
  for b in 4224 5632 7168 8176 10224 12288; do
    l=$(timeout -k 10 120 python3 bench.py --config $c --sampling bilinear --budget $b --steps 40 --warmup 5 --no-cpu-baseline --no-configs 2>>$O/err.log | tail -1)
    echo "$c budget $b $(echo "$l" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); p=d["roofline"]["plan"]; print(d["roofline"]["kernel_ms_per_frame"], "lean", p["lean_tiles"], "direct", p["direct_tiles"])')" | tee -a $O/budget.log
  done
done
