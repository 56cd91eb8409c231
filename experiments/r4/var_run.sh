#!/bin/bash
# bilinear kernel time per config under each experiments/r4/lib*.so ($1 = tag, rest = configs)
R=$GRAFT_REPO_ROOT; T=$1; shift; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
for lib in $(ls experiments/r4/lib*.so); do
  for c in "$@"; do
    l=$(PB_LIB_PATH=$R/$lib timeout -k 10 120 python3 bench.py --config $c --sampling bilinear --steps 40 --warmup 5 --no-cpu-baseline --no-configs 2>>$O/err.log | tail -1)
    echo "$(basename $lib) $c $(echo "$l" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms_per_frame"])')" | tee -a $O/var.log
  done
done
